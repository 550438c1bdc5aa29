/*
 * yn_oracle.c — CPU restatement of the YOLO-Nano hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (yolo-nano_amd/)
 * never links or calls it and fails loudly when the HIP library is missing.
 *
 * Parity pinning: every function below is checked against fixtures produced by
 * importing the reference itself (tests/golden/gen_golden.py -> tests/golden/*.npz;
 * tests/test_oracle_golden.py).  The reference has no tests or golden vectors of
 * its own, so those generated fixtures are the pin.
 *
 * Layout follows the reference: float32, NCHW.  Each function cites the
 * reference lines it restates (paths relative to /root/reference).
 *
 * Built by oracle/build.py with  gcc -O2 -fopenmp -ffp-contract=off : every float
 * operation is a single IEEE-754 binary32 operation in source order, which is what
 * numpy does for the NMS arithmetic that must match bit for bit.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define YO_API __attribute__((visibility("default")))

YO_API int yo_version(void) { return 1; }

YO_API int yo_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

YO_API void yo_set_num_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

/* ---------------------------------------------------------------------------------------------
 * nn.Conv2d, square kernel k, stride s, zero padding p, groups g, optional bias.
 * Used for: stem (backbone/shufflenetv2.py:109), depthwise (:65-67), pointwise (:45,54,59),
 * neck/head convs (utils/modules.py:12, models/yolo_nano.py:40-70).
 * Accumulation: float32, order (ci, ky, kx), bias added first.
 * ------------------------------------------------------------------------------------------- */
YO_API void yo_conv2d(const float* x, int B, int Cin, int H, int W,
                      const float* w, const float* bias, int Cout, int k, int stride, int pad, int groups,
                      float* y)
{
    const int Ho = (H + 2 * pad - k) / stride + 1;
    const int Wo = (W + 2 * pad - k) / stride + 1;
    const int cig = Cin / groups, cog = Cout / groups;
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b) {
        for (int co = 0; co < Cout; ++co) {
            const int g = co / cog;
            float* yo = y + ((size_t)b * Cout + co) * Ho * Wo;
            const float bv = bias ? bias[co] : 0.0f;
            for (int i = 0; i < Ho * Wo; ++i) yo[i] = bv;
            for (int ci = 0; ci < cig; ++ci) {
                const float* xi = x + ((size_t)b * Cin + g * cig + ci) * H * W;
                const float* wk = w + ((size_t)co * cig + ci) * k * k;
                for (int ky = 0; ky < k; ++ky) {
                    for (int kx = 0; kx < k; ++kx) {
                        const float wv = wk[ky * k + kx];
                        for (int oy = 0; oy < Ho; ++oy) {
                            const int iy = oy * stride - pad + ky;
                            if (iy < 0 || iy >= H) continue;
                            /* ox range with 0 <= ox*stride - pad + kx < W */
                            int ox0 = 0;
                            while (ox0 < Wo && ox0 * stride - pad + kx < 0) ++ox0;
                            int ox1 = Wo;
                            while (ox1 > ox0 && (ox1 - 1) * stride - pad + kx >= W) --ox1;
                            const float* xr = xi + (size_t)iy * W - pad + kx;
                            float* yr = yo + (size_t)oy * Wo;
                            if (stride == 1) {
                                for (int ox = ox0; ox < ox1; ++ox) yr[ox] += wv * xr[ox];
                            } else {
                                for (int ox = ox0; ox < ox1; ++ox) yr[ox] += wv * xr[ox * stride];
                            }
                        }
                    }
                }
            }
        }
    }
}

/* nn.BatchNorm2d in eval mode: y = (x - mean) / sqrt(var + eps) * gamma + beta  (per channel). */
YO_API void yo_bn_eval(float* x, int B, int C, int HW, const float* gamma, const float* beta,
                       const float* mean, const float* var, float eps)
{
#pragma omp parallel for collapse(2) schedule(static)
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c) {
            float* p = x + ((size_t)b * C + c) * HW;
            const float inv = 1.0f / sqrtf(var[c] + eps);
            for (int i = 0; i < HW; ++i) p[i] = (p[i] - mean[c]) * inv * gamma[c] + beta[c];
        }
}

/* act: 0 none, 1 ReLU (backbone/shufflenetv2.py:48,57,62,112), 2 LeakyReLU(0.1) (utils/modules.py:14) */
YO_API void yo_act(float* x, size_t n, int act)
{
    if (act == 1) { for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.0f ? x[i] : 0.0f; }
    else if (act == 2) { for (size_t i = 0; i < n; ++i) x[i] = x[i] > 0.0f ? x[i] : 0.1f * x[i]; }
}

/* utils/fuse_conv_bn.py:6-22 : f = gamma/sqrt(var+eps); W' = W*f[co]; b' = (b - mean)*f + beta */
YO_API void yo_fold_conv_bn(const float* w, const float* b, int Cout, int per_out,
                            const float* gamma, const float* beta, const float* mean, const float* var, float eps,
                            float* w_out, float* b_out)
{
    for (int co = 0; co < Cout; ++co) {
        const float f = gamma[co] / sqrtf(var[co] + eps);
        for (int i = 0; i < per_out; ++i) w_out[(size_t)co * per_out + i] = w[(size_t)co * per_out + i] * f;
        const float bv = b ? b[co] : 0.0f;
        b_out[co] = (bv - mean[co]) * f + beta[co];
    }
}

/* nn.MaxPool2d(kernel_size=3, stride=2, padding=1) — backbone/shufflenetv2.py:116 (implicit -inf pad) */
YO_API void yo_maxpool3x3s2(const float* x, int B, int C, int H, int W, float* y)
{
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
#pragma omp parallel for schedule(static)
    for (int bc = 0; bc < B * C; ++bc) {
        const float* xi = x + (size_t)bc * H * W;
        float* yo = y + (size_t)bc * Ho * Wo;
        for (int oy = 0; oy < Ho; ++oy)
            for (int ox = 0; ox < Wo; ++ox) {
                float m = -INFINITY;
                for (int ky = 0; ky < 3; ++ky) {
                    const int iy = oy * 2 - 1 + ky;
                    if (iy < 0 || iy >= H) continue;
                    for (int kx = 0; kx < 3; ++kx) {
                        const int ix = ox * 2 - 1 + kx;
                        if (ix < 0 || ix >= W) continue;
                        const float v = xi[iy * W + ix];
                        if (v > m) m = v;
                    }
                }
                yo[oy * Wo + ox] = m;
            }
    }
}

/* backbone/shufflenetv2.py:14-28 : view(B,g,C/g,H,W).transpose(1,2) -> out[:, j*g + gi] = in[:, gi*(C/g) + j] */
YO_API void yo_channel_shuffle(const float* x, int B, int C, int HW, int groups, float* y)
{
    const int cpg = C / groups;
    for (int b = 0; b < B; ++b)
        for (int gi = 0; gi < groups; ++gi)
            for (int j = 0; j < cpg; ++j)
                memcpy(y + ((size_t)b * C + j * groups + gi) * HW, x + ((size_t)b * C + gi * cpg + j) * HW, sizeof(float) * HW);
}

/* models/yolo_nano.py:291-292 : a + F.interpolate(b, scale_factor=2.0)  (nearest: src = dst/2) */
YO_API void yo_add_up2(const float* a, const float* b, int BC, int H, int W, float* y)
{
    const int h2 = H / 2, w2 = W / 2;
    for (int c = 0; c < BC; ++c)
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < W; ++xx)
                y[((size_t)c * H + yy) * W + xx] = a[((size_t)c * H + yy) * W + xx] + b[((size_t)c * h2 + yy / 2) * w2 + xx / 2];
}

/* models/yolo_nano.py:295-296 : a + F.interpolate(b, scale_factor=0.5)  (nearest: src = 2*dst) */
YO_API void yo_add_down2(const float* a, const float* b, int BC, int H, int W, float* y)
{
    const int h2 = H * 2, w2 = W * 2;
    for (int c = 0; c < BC; ++c)
        for (int yy = 0; yy < H; ++yy)
            for (int xx = 0; xx < W; ++xx)
                y[((size_t)c * H + yy) * W + xx] = a[((size_t)c * H + yy) * W + xx] + b[((size_t)c * h2 + 2 * yy) * w2 + 2 * xx];
}

/* ---------------------------------------------------------------------------------------------
 * models/yolo_nano.py:308-330 (head re-layout) + :120-156 (decode_xywh/decode_boxes) + :365-367.
 * heads[s] is the raw NCHW head tensor of ONE image: [A*(1+C+4), Hs, Ws].
 * Channel map: obj = a ; cls = A + a*C + c ; box = A*(1+C) + a*4 + k.
 * Candidate n = off_s + (y*Ws + x)*A + a.
 * Outputs: all_bbox [N,4] = clamp(xyxy/S, 0, 1) ; all_class [N,C] = softmax(cls)*sigmoid(obj).
 * ------------------------------------------------------------------------------------------- */
static float sigmoidf_(float v) { return 1.0f / (1.0f + expf(-v)); }

YO_API void yo_score_decode(const float* const* heads, int S, int C, int A, const float* anchors /* [3][A][2] */,
                            float* all_bbox, float* all_class)
{
    static const int strides[3] = {8, 16, 32};
    int off = 0;
    for (int s = 0; s < 3; ++s) {
        const int Ws = S / strides[s], Hs = S / strides[s], HW = Hs * Ws;
        const float* h = heads[s];
        for (int cell = 0; cell < HW; ++cell) {
            const int gy = cell / Ws, gx = cell % Ws;
            for (int a = 0; a < A; ++a) {
                const int n = off + cell * A + a;
                const float obj = sigmoidf_(h[(size_t)a * HW + cell]);
                /* softmax over classes (torch.softmax: subtract max, exp, divide by sum) */
                float mx = -INFINITY;
                for (int c = 0; c < C; ++c) { float v = h[(size_t)(A + a * C + c) * HW + cell]; if (v > mx) mx = v; }
                float sum = 0.0f;
                for (int c = 0; c < C; ++c) { float e = expf(h[(size_t)(A + a * C + c) * HW + cell] - mx); all_class[(size_t)n * C + c] = e; sum += e; }
                for (int c = 0; c < C; ++c) all_class[(size_t)n * C + c] = all_class[(size_t)n * C + c] / sum * obj;
                const float* t = h + (size_t)(A * (1 + C) + a * 4) * HW + cell;
                const float tx = t[0], ty = t[(size_t)HW], tw = t[(size_t)2 * HW], th = t[(size_t)3 * HW];
                const float cx = (sigmoidf_(tx) + (float)gx) * (float)strides[s];
                const float cy = (sigmoidf_(ty) + (float)gy) * (float)strides[s];
                const float bw = expf(tw) * anchors[(s * A + a) * 2 + 0];
                const float bh = expf(th) * anchors[(s * A + a) * 2 + 1];
                float bx[4] = {cx - bw / 2, cy - bh / 2, cx + bw / 2, cy + bh / 2};
                for (int k = 0; k < 4; ++k) {
                    float v = bx[k] / (float)S;
                    v = v < 0.0f ? 0.0f : (v > 1.0f ? 1.0f : v);
                    all_bbox[(size_t)n * 4 + k] = v;
                }
            }
        }
        off += HW * A;
    }
}

/* models/yolo_nano.py:139-156 on a [HW_total, A, 4] txtytwth tensor of one image -> xyxy pixels [N,4] */
YO_API void yo_decode_boxes(const float* txtytwth, int S, int A, const float* anchors, float* xywh, float* xyxy)
{
    static const int strides[3] = {8, 16, 32};
    int n = 0;
    for (int s = 0; s < 3; ++s) {
        const int Ws = S / strides[s], HW = Ws * Ws;
        for (int cell = 0; cell < HW; ++cell) {
            const int gy = cell / Ws, gx = cell % Ws;
            for (int a = 0; a < A; ++a, ++n) {
                const float* t = txtytwth + (size_t)n * 4;
                const float cx = (sigmoidf_(t[0]) + (float)gx) * (float)strides[s];
                const float cy = (sigmoidf_(t[1]) + (float)gy) * (float)strides[s];
                const float bw = expf(t[2]) * anchors[(s * A + a) * 2 + 0];
                const float bh = expf(t[3]) * anchors[(s * A + a) * 2 + 1];
                if (xywh) { xywh[n * 4 + 0] = cx; xywh[n * 4 + 1] = cy; xywh[n * 4 + 2] = bw; xywh[n * 4 + 3] = bh; }
                xyxy[n * 4 + 0] = cx - bw / 2; xyxy[n * 4 + 1] = cy - bh / 2;
                xyxy[n * 4 + 2] = cx + bw / 2; xyxy[n * 4 + 3] = cy + bh / 2;
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * models/yolo_nano.py:159-188 (nms), :191-242 (diou_nms), utils/misc.py:8-37 (nms with thresh arg).
 * float32 arithmetic exactly as numpy evaluates it; survivors are those with `ovr <= thresh`
 * (NaN -> dropped).  Order = scores.argsort()[::-1]; numpy's default sort is unstable, so the tie
 * order is pinned HERE as: equal scores -> higher index first (== argsort(kind='stable')[::-1]).
 * Returns the number of kept boxes; keep[] gets indices in pick order.
 * ------------------------------------------------------------------------------------------- */
typedef struct { float s; int i; } yo_si;
static int yo_cmp_desc(const void* a, const void* b)
{
    const yo_si* p = (const yo_si*)a; const yo_si* q = (const yo_si*)b;
    if (p->s > q->s) return -1;
    if (p->s < q->s) return 1;
    return (p->i > q->i) ? -1 : (p->i < q->i);
}

YO_API int yo_nms(const float* dets, const float* scores, int n, float thresh, int diou, int64_t* keep)
{
    if (n <= 0) return 0;
    yo_si* ord = (yo_si*)malloc(sizeof(yo_si) * n);
    float* areas = (float*)malloc(sizeof(float) * n);
    int* order = (int*)malloc(sizeof(int) * n);
    for (int i = 0; i < n; ++i) {
        ord[i].s = scores[i]; ord[i].i = i;
        areas[i] = (dets[i * 4 + 2] - dets[i * 4 + 0]) * (dets[i * 4 + 3] - dets[i * 4 + 1]);
    }
    qsort(ord, n, sizeof(yo_si), yo_cmp_desc);
    for (int i = 0; i < n; ++i) order[i] = ord[i].i;
    int m = n, nk = 0;
    while (m > 0) {
        const int i = order[0];
        keep[nk++] = i;
        const float x1i = dets[i * 4], y1i = dets[i * 4 + 1], x2i = dets[i * 4 + 2], y2i = dets[i * 4 + 3];
        int w = 0;
        for (int t = 1; t < m; ++t) {
            const int j = order[t];
            const float x1j = dets[j * 4], y1j = dets[j * 4 + 1], x2j = dets[j * 4 + 2], y2j = dets[j * 4 + 3];
            const float xx1 = x1i > x1j ? x1i : x1j;       /* np.maximum */
            const float yy1 = y1i > y1j ? y1i : y1j;
            const float xx2 = x2i < x2j ? x2i : x2j;       /* np.minimum */
            const float yy2 = y2i < y2j ? y2i : y2j;
            float ww = xx2 - xx1; ww = ww > 1e-28f ? ww : 1e-28f;
            float hh = yy2 - yy1; hh = hh > 1e-28f ? hh : 1e-28f;
            const float inter = ww * hh;
            const float t0 = areas[i] + areas[j];
            float ovr = inter / (t0 - inter);
            if (diou) {
                /* models/yolo_nano.py:216-236 */
                float mxx = x1i, mnx = x1i, mxy = y1i, mny = y1i;
                const float xs[3] = {x2i, x1j, x2j}, ys[3] = {y2i, y1j, y2j};
                for (int q = 0; q < 3; ++q) { if (xs[q] > mxx) mxx = xs[q]; if (xs[q] < mnx) mnx = xs[q]; if (ys[q] > mxy) mxy = ys[q]; if (ys[q] < mny) mny = ys[q]; }
                const float dx = mxx - mnx, dy = mxy - mny;
                const float Cd = sqrtf(dx * dx + dy * dy);
                const float p1x = (x1i + x2i) / 2.0f, p1y = (y1i + y2i) / 2.0f;
                const float p2x = (x1j + x2j) / 2.0f, p2y = (y1j + y2j) / 2.0f;
                const float ex = p2x - p1x, ey = p2y - p1y;
                const float D = sqrtf(ex * ex + ey * ey);
                const float lens = (D * D) / (Cd * Cd + 1e-20f);
                ovr = ovr - lens;
            }
            if (ovr <= thresh) order[w++] = j;
        }
        m = w;
    }
    free(ord); free(areas); free(order);
    return nk;
}

/* ---------------------------------------------------------------------------------------------
 * models/yolo_nano.py:245-279 postprocess: argmax over classes (first max), score gather,
 * keep score >= conf_thresh, per-class NMS, result = flagged candidates in ascending index order.
 * Returns K; out_idx[K] are candidate indices into the N inputs.
 * ------------------------------------------------------------------------------------------- */
YO_API int yo_postprocess(const float* all_local, const float* all_conf, int N, int C,
                          float conf_thresh, float nms_thresh, int diou,
                          float* out_boxes, float* out_scores, int64_t* out_cls, int64_t* out_idx)
{
    int* cls = (int*)malloc(sizeof(int) * (N > 0 ? N : 1));
    float* sc = (float*)malloc(sizeof(float) * (N > 0 ? N : 1));
    int* cand = (int*)malloc(sizeof(int) * (N > 0 ? N : 1));
    int M = 0;
    for (int n = 0; n < N; ++n) {
        int best = 0; float bv = all_conf[(size_t)n * C];
        for (int c = 1; c < C; ++c) { const float v = all_conf[(size_t)n * C + c]; if (v > bv) { bv = v; best = c; } }
        if (bv >= conf_thresh) { cand[M] = n; cls[M] = best; sc[M] = bv; ++M; }
    }
    unsigned char* flag = (unsigned char*)calloc(M > 0 ? M : 1, 1);
    float* cb = (float*)malloc(sizeof(float) * 4 * (M > 0 ? M : 1));
    float* cs = (float*)malloc(sizeof(float) * (M > 0 ? M : 1));
    int* ci = (int*)malloc(sizeof(int) * (M > 0 ? M : 1));
    int64_t* ck = (int64_t*)malloc(sizeof(int64_t) * (M > 0 ? M : 1));
    for (int c = 0; c < C; ++c) {
        int m = 0;
        for (int t = 0; t < M; ++t) if (cls[t] == c) { memcpy(cb + 4 * m, all_local + (size_t)cand[t] * 4, 16); cs[m] = sc[t]; ci[m] = t; ++m; }
        if (!m) continue;
        const int nk = yo_nms(cb, cs, m, nms_thresh, diou, ck);
        for (int q = 0; q < nk; ++q) flag[ci[ck[q]]] = 1;
    }
    int K = 0;
    for (int t = 0; t < M; ++t) if (flag[t]) {
        memcpy(out_boxes + 4 * K, all_local + (size_t)cand[t] * 4, 16);
        out_scores[K] = sc[t]; out_cls[K] = cls[t]; if (out_idx) out_idx[K] = cand[t]; ++K;
    }
    free(cls); free(sc); free(cand); free(flag); free(cb); free(cs); free(ci); free(ck);
    return K;
}

// kernels_train.hip — training-side kernels of the YOLO-Nano hot path.
//
//   loss_kernel<HEADS>  models/yolo_nano.py:332-358 in one pass over the candidates: box decode (/S, unclamped),
//                       tools.iou_score (tools.py:219-233), label assembly (gt_conf = iou.detach()), tools.loss
//                       (tools.py:236-276: objectness MSE-with-logits, class cross-entropy, txty BCE + twth MSE
//                       weighted+masked, SmoothL1 on iou; every term sum/B) AND the gradient of the SUM of the four
//                       losses (train.py:222) w.r.t. the raw predictions.  HEADS = false: predictions in the
//                       reference's split layout conf [B,N], cls [B,N,C], txtytwth [B,N,4]; HEADS = true: read
//                       from / write gradients to the three raw NHWC head tensors directly (no re-layout pass).
//   loss_reduce_kernel  deterministic final sum of the per-block partials.
#include "yn_internal.h"
#include "yn_h16.h"

namespace ynk {

__device__ __forceinline__ float sigmoid_t(float v) { return 1.0f / (1.0f + expf(-v)); }

struct LossArgs {
    const float* conf; const float* cls; const float* t;          // split layout (HEADS = false)
    const void* head[3]; void* ghead[3];                          // head layout  (HEADS = true): HT = float or _Float16
    const float* gscale;                                          // device pointer to the loss scale the gradients are multiplied by, or null
    const float* target;                                          // [B,N,11] = obj, cls, tx,ty,tw,th, weight, x1,y1,x2,y2
    float* g_conf; float* g_cls; float* g_t;                      // may be null (forward only)
    float* partial;                                               // [gridDim.x][4]
    GridInfo g;
    int B;
};

template <bool HEADS, typename HT = float>
__global__ __launch_bounds__(256) void loss_kernel(LossArgs a)
{
    __shared__ float red[4][4];
    const GridInfo& g = a.g;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    float l_conf = 0.0f, l_cls = 0.0f, l_box = 0.0f, l_iou = 0.0f;
    if (i < (long)a.B * g.N) {
        const int b = (int)(i / g.N), n = (int)(i - (long)b * g.N);
        const int s = (n >= g.off[2]) ? 2 : ((n >= g.off[1]) ? 1 : 0);
        const int local = n - g.off[s];
        const int cell = local / g.A, an = local - cell * g.A;
        const int gy = cell / g.w[s], gx = cell - gy * g.w[s];
        const float stride = (float)(8 << s), S = (float)g.S, invB = 1.0f / (float)a.B;
        const int HC = g.head_ld;
        const HT* pconf; const HT* pcls; const HT* pt;
        HT *qconf = nullptr, *qcls = nullptr, *qt = nullptr;
        const float gs = a.gscale ? a.gscale[0] : 1.0f;           // loss scale (fp16 step): every gradient written below carries it
        if (HEADS) {
            const size_t row = ((size_t)b * g.hw[s] + cell) * HC;
            const HT* hp = reinterpret_cast<const HT*>(a.head[s]);
            HT* gp = reinterpret_cast<HT*>(a.ghead[s]);
            pconf = hp + row + an; pcls = hp + row + g.A + an * g.C; pt = hp + row + g.A * (1 + g.C) + an * 4;
            if (gp) { qconf = gp + row + an; qcls = gp + row + g.A + an * g.C; qt = gp + row + g.A * (1 + g.C) + an * 4; }
        } else {
            pconf = reinterpret_cast<const HT*>(a.conf) + i; pcls = reinterpret_cast<const HT*>(a.cls) + (size_t)i * g.C; pt = reinterpret_cast<const HT*>(a.t) + (size_t)i * 4;
            if (a.g_conf) { qconf = reinterpret_cast<HT*>(a.g_conf) + i; qcls = reinterpret_cast<HT*>(a.g_cls) + (size_t)i * g.C; qt = reinterpret_cast<HT*>(a.g_t) + (size_t)i * 4; }
        }
        const float* tg = a.target + (size_t)i * 11;
        const float obj = tg[0], wgt = tg[6];
        const int gcls = (int)tg[1];
        const float pos = obj == 1.0f ? 1.0f : 0.0f, neg = obj == 0.0f ? 1.0f : 0.0f, mask = obj > 0.0f ? 1.0f : 0.0f;
        // decode (models/yolo_nano.py:120-156) / S, unclamped
        const float tx = (float)pt[0], ty = (float)pt[1], tw = (float)pt[2], th = (float)pt[3];
        const float sx = sigmoid_t(tx), sy = sigmoid_t(ty), ew = expf(tw), eh = expf(th);
        const float aw = g.anchors[(s * g.A + an) * 2], ah = g.anchors[(s * g.A + an) * 2 + 1];
        const float cx = (sx + (float)gx) * stride, cy = (sy + (float)gy) * stride, bw = ew * aw, bh = eh * ah;
        const float ax1 = (cx - bw / 2) / S, ay1 = (cy - bh / 2) / S, ax2 = (cx + bw / 2) / S, ay2 = (cy + bh / 2) / S;
        const float bx1 = tg[7], by1 = tg[8], bx2 = tg[9], by2 = tg[10];
        // tools.iou_score
        const float tlx = fmaxf(ax1, bx1), tly = fmaxf(ay1, by1), brx = fminf(ax2, bx2), bry = fminf(ay2, by2);
        const float wa = ax2 - ax1, ha = ay2 - ay1;
        const float Aa = wa * ha, Ab = (bx2 - bx1) * (by2 - by1);
        const float en = (tlx < brx && tly < bry) ? 1.0f : 0.0f;
        const float wi = brx - tlx, hi = bry - tly;
        const float I = wi * hi * en;
        const float U = Aa + Ab - I;
        const float iou = I / U;
        // objectness (tools.py:12-34); gt_conf = iou.detach()
        const float sg = sigmoid_t((float)pconf[0]);
        l_conf = (5.0f * pos * (sg - iou) * (sg - iou) + neg * sg * sg) * invB;
        if (qconf) qconf[0] = (HT)((5.0f * pos * 2.0f * (sg - iou) + neg * 2.0f * sg) * sg * (1.0f - sg) * invB * gs);
        // class cross-entropy * mask
        // candidates that are not positives contribute nothing to the class term: their (pre-zeroed, see the callers)
        // gradient slots are left untouched, so only the handful of positives walk the class vector
        if (mask > 0.0f) {
            float mx = -INFINITY;
            for (int c = 0; c < g.C; ++c) mx = fmaxf(mx, (float)pcls[c]);
            float sum = 0.0f;
            for (int c = 0; c < g.C; ++c) sum += expf((float)pcls[c] - mx);
            if (mask > 0.0f) l_cls = (mx + logf(sum) - (float)pcls[gcls]) * invB;
            if (qcls) {
                const float k = mask * invB / sum;
                for (int c = 0; c < g.C; ++c) {
                    float gr = expf((float)pcls[c] - mx) * k;
                    if (c == gcls) gr -= mask * invB;
                    qcls[c] = (HT)(gr * gs);
                }
            }
        }
        // box regression: BCE-with-logits on txty, MSE on twth, both * weight * mask
        const float wm = wgt * mask * invB;
        {
            const float g0 = tg[2], g1 = tg[3], g2 = tg[4], g3 = tg[5];
            const float b0 = fmaxf(tx, 0.0f) - tx * g0 + log1pf(expf(-fabsf(tx)));
            const float b1 = fmaxf(ty, 0.0f) - ty * g1 + log1pf(expf(-fabsf(ty)));
            l_box = ((b0 + b1) + ((tw - g2) * (tw - g2) + (th - g3) * (th - g3))) * wm;
            // SmoothL1(iou, mask)  (beta = 1), over every candidate
            const float d = iou - mask;
            l_iou = (fabsf(d) < 1.0f ? 0.5f * d * d : fabsf(d) - 0.5f) * invB;
            if (qt) {
                float gt0 = (sx - g0) * wm, gt1 = (sy - g1) * wm, gt2 = 2.0f * (tw - g2) * wm, gt3 = 2.0f * (th - g3) * wm;
                // d iou_loss / d box, through iou = I / (Aa + Ab - I); torch.max/min split the gradient on ties
                const float giou = (fabsf(d) < 1.0f ? d : (d > 0.0f ? 1.0f : -1.0f)) * invB;
                const float wx1 = ax1 > bx1 ? 1.0f : (ax1 == bx1 ? 0.5f : 0.0f), wy1 = ay1 > by1 ? 1.0f : (ay1 == by1 ? 0.5f : 0.0f);
                const float wx2 = ax2 < bx2 ? 1.0f : (ax2 == bx2 ? 0.5f : 0.0f), wy2 = ay2 < by2 ? 1.0f : (ay2 == by2 ? 0.5f : 0.0f);
                const float dI[4] = {-hi * en * wx1, -wi * en * wy1, hi * en * wx2, wi * en * wy2};
                const float dA[4] = {-ha, -wa, ha, wa};
                float gb[4];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float v = (dI[k] * U - I * (dA[k] - dI[k])) / (U * U);
                    if (!(v == v) || fabsf(v) == INFINITY) v = 0.0f;      // 0/0 for degenerate pairs carries no gradient
                    gb[k] = v * giou;
                }
                const float dcx = sx * (1.0f - sx) * stride / S, dcy = sy * (1.0f - sy) * stride / S;
                gt0 += (gb[0] + gb[2]) * dcx;
                gt1 += (gb[1] + gb[3]) * dcy;
                gt2 += 0.5f * (gb[2] - gb[0]) * (bw / S);
                gt3 += 0.5f * (gb[3] - gb[1]) * (bh / S);
                qt[0] = (HT)(gt0 * gs); qt[1] = (HT)(gt1 * gs); qt[2] = (HT)(gt2 * gs); qt[3] = (HT)(gt3 * gs);
            }
        }
    }
    // deterministic block reduction: wave shuffle, then 4 waves through LDS
    float v[4] = {l_conf, l_cls, l_box, l_iou};
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { red[wave][0] = v[0]; red[wave][1] = v[1]; red[wave][2] = v[2]; red[wave][3] = v[3]; }
    __syncthreads();
    if (threadIdx.x < 4) a.partial[(size_t)blockIdx.x * 4 + threadIdx.x] = (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ partial, int nblocks, float* __restrict__ out)
{
    __shared__ double red[4][4];
    double v[4] = {0.0, 0.0, 0.0, 0.0};
    for (int i = threadIdx.x; i < nblocks; i += 256)
#pragma unroll
        for (int k = 0; k < 4; ++k) v[k] += (double)partial[(size_t)i * 4 + k];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[k] += __shfl_xor(v[k], off);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) for (int k = 0; k < 4; ++k) red[wave][k] = v[k];
    __syncthreads();
    if (threadIdx.x < 4) out[threadIdx.x] = (float)((red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]));
}

int loss_num_blocks(const GridInfo& g, int B) { return (int)(((long)B * g.N + 255) / 256); }

void launch_loss(const float* conf, const float* cls, const float* t, const float* const head[3], float* const ghead[3],
                 const float* target, const GridInfo& g, int B, float* partial, float* losses,
                 float* g_conf, float* g_cls, float* g_t, hipStream_t s)
{
    LossArgs a{};
    a.conf = conf; a.cls = cls; a.t = t; a.target = target; a.g_conf = g_conf; a.g_cls = g_cls; a.g_t = g_t;
    a.partial = partial; a.g = g; a.B = B;
    const int nb = loss_num_blocks(g, B);
    if (head) {
        for (int k = 0; k < 3; ++k) { a.head[k] = head[k]; a.ghead[k] = ghead ? ghead[k] : nullptr; }
        hipLaunchKernelGGL((loss_kernel<true, float>), dim3(nb), dim3(256), 0, s, a);
    } else {
        hipLaunchKernelGGL((loss_kernel<false, float>), dim3(nb), dim3(256), 0, s, a);
    }
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, partial, nb, losses);
}

// the same pass over fp16 head tensors (the fp16 training step): losses in fp32, gradients * loss scale stored as fp16
void launch_loss_h16(const h16* const head[3], h16* const ghead[3], const float* target, const GridInfo& g, int B, float* partial, float* losses,
                     const float* scale_state, hipStream_t s)
{
    LossArgs a{};
    a.target = target; a.partial = partial; a.g = g; a.B = B; a.gscale = scale_state;
    for (int k = 0; k < 3; ++k) { a.head[k] = head[k]; a.ghead[k] = ghead ? ghead[k] : nullptr; }
    const int nb = loss_num_blocks(g, B);
    hipLaunchKernelGGL((loss_kernel<true, h16>), dim3(nb), dim3(256), 0, s, a);
    hipLaunchKernelGGL(loss_reduce_kernel, dim3(1), dim3(256), 0, s, partial, nb, losses);
}

// -------------------------------------------------------------------------------------------------
// Training label assigner: tools.multi_gt_creator (tools.py:97-216) with compute_iou (tools.py:36-76).
// The reference walks the objects of an image IN LIST ORDER and later objects overwrite the (cell, anchor) slot of
// earlier ones (an 'ignore' write only touches obj and weight), so one thread owns one image and keeps that order;
// float64 arithmetic like numpy, float32 at the store.  `target` must be zero-filled by the caller.
// -------------------------------------------------------------------------------------------------
struct TargetArgs {
    const double* labels;          // [total][5] xmin, ymin, xmax, ymax, class
    const int32_t* offsets;        // [B+1]
    float* target;                 // [B][N][11]
    double anchors[18];
    int B, S, A, N;
    int w[3], off[3];
};

// One workgroup per image.  The float64 arithmetic of an object (nine anchor IoUs, two logarithms, the divisions) does not depend on any other
// object, so the image's objects are evaluated side by side, 64 at a time, each into its own list of (slot, 11 values | ignore) records in LDS;
// ONE thread then replays the records in list order, which is all the reference's overwrite rule needs (the first form ran the whole chain
// serially in one thread per image: 68 us per step for 32 images).
__global__ __launch_bounds__(64) void targets_kernel(TargetArgs a)
{
    __shared__ int r_n[64];
    __shared__ int r_slot[64][9];
    __shared__ signed char r_pos[64][9];
    __shared__ float r_val[64][10];                         // the positive's t[1..10] (an object has at most one positive: its best anchor)
    const int b = blockIdx.x;
    if (b >= a.B) return;
    const double w = (double)a.S, h = (double)a.S;
    const int na = 3 * a.A;
    float* out = a.target + (size_t)b * a.N * 11;
    const int l0 = a.offsets[b], l1 = a.offsets[b + 1];
    for (int base = l0; base < l1; base += 64) {
        const int li = base + (int)threadIdx.x;
        int cnt = 0;
        if (li < l1) {
            const double* lab = a.labels + (size_t)li * 5;
            const double xmin = lab[0], ymin = lab[1], xmax = lab[2], ymax = lab[3];
            const int cls = (int)lab[4];
            const double c_x = (xmax + xmin) / 2 * w, c_y = (ymax + ymin) / 2 * h;
            const double box_w = (xmax - xmin) * w, box_h = (ymax - ymin) * h;
            if (!(box_w < 1. || box_h < 1.)) {                                       // tools.py:122-124
                double iou[9];
                int best = 0;
                bool any_above = false;
                for (int i = 0; i < na; ++i) {
                    const double aw = a.anchors[2 * i], ah = a.anchors[2 * i + 1];
                    const double ax1 = 0.0 - aw / 2, ay1 = 0.0 - ah / 2, ax2 = 0.0 + aw / 2, ay2 = 0.0 + ah / 2;
                    const double gx1 = 0.0 - box_w / 2, gy1 = 0.0 - box_h / 2, gx2 = 0.0 + box_w / 2, gy2 = 0.0 + box_h / 2;
                    const double i_w = fmin(gx2, ax2) - fmax(gx1, ax1);
                    const double i_h = fmin(gy2, ay2) - fmax(gy1, ay1);
                    const double s_i = i_h * i_w;
                    const double u = box_w * box_h + aw * ah - s_i + 1e-20;
                    iou[i] = s_i / u;
                    if (iou[i] > iou[best]) best = i;                                   // np.argmax: first maximum
                    any_above |= iou[i] > 0.5;
                }
                for (int index = 0; index < na; ++index) {
                    const bool is_best = index == best;
                    if (any_above ? !(iou[index] > 0.5) : !is_best) continue;
                    const int si = index / a.A, ab = index - si * a.A;
                    const double s = (double)(8 << si);
                    const double c_x_s = c_x / s, c_y_s = c_y / s;
                    const int gx = (int)c_x_s, gy = (int)c_y_s;
                    if (gx < 0 || gy < 0 || gx >= a.w[si] || gy >= a.w[si]) continue; // positives: tools.py:157; ignore writes would raise in numpy
                    r_slot[threadIdx.x][cnt] = a.off[si] + (gy * a.w[si] + gx) * a.A + ab;
                    r_pos[threadIdx.x][cnt] = is_best ? 1 : 0;
                    if (is_best) {
                        const double pw = a.anchors[2 * index], ph = a.anchors[2 * index + 1];
                        float* t = r_val[threadIdx.x];
                        t[0] = (float)cls;
                        t[1] = (float)(c_x_s - gx); t[2] = (float)(c_y_s - gy);
                        t[3] = (float)log(box_w / pw); t[4] = (float)log(box_h / ph);
                        t[5] = (float)(2.0 - (box_w / w) * (box_h / h));
                        t[6] = (float)xmin; t[7] = (float)ymin; t[8] = (float)xmax; t[9] = (float)ymax;
                    }
                    ++cnt;
                }
            }
        }
        r_n[threadIdx.x] = cnt;
        __syncthreads();
        if (threadIdx.x == 0) {                                                      // the reference's order: object by object, anchor index ascending
            const int m = l1 - base < 64 ? l1 - base : 64;
            for (int o = 0; o < m; ++o)
                for (int k = 0; k < r_n[o]; ++k) {
                    float* t = out + (size_t)r_slot[o][k] * 11;
                    if (r_pos[o][k]) {
                        t[0] = 1.0f;
#pragma unroll
                        for (int q = 0; q < 10; ++q) t[1 + q] = r_val[o][q];
                    } else {
                        t[0] = -1.0f;                                                   // tools.py:206-207
                        t[6] = -1.0f;
                    }
                }
        }
        __syncthreads();
    }
}

void launch_make_targets(const double* labels, const int32_t* offsets, int B, const double* anchors18, const GridInfo& g, float* target, hipStream_t s)
{
    TargetArgs a{};
    a.labels = labels; a.offsets = offsets; a.target = target; a.B = B; a.S = g.S; a.A = g.A; a.N = g.N;
    for (int i = 0; i < 18; ++i) a.anchors[i] = i < 6 * g.A ? anchors18[i] : 0.0;
    for (int k = 0; k < 3; ++k) { a.w[k] = g.w[k]; a.off[k] = g.off[k]; }
    hipLaunchKernelGGL(targets_kernel, dim3(B), dim3(64), 0, s, a);
}

// -------------------------------------------------------------------------------------------------
// torch.optim.SGD(momentum, weight_decay) step (train.py:167-171, 230) over one FLAT parameter bucket, fused with
// the 1/world_size scaling of the all-reduced gradient sum (one flat bucket, one RCCL all-reduce per step):
//     g = grad * grad_scale + wd * p ;  buf = first ? g : momentum * buf + g ;  p -= lr * buf
// (dampening 0, nesterov off — the reference's settings).  16-byte accesses, grid-stride.
// -------------------------------------------------------------------------------------------------
// flag[0] = 1 when the gradient bucket holds a NaN / Inf (zeroed by the launcher first): the step is then skipped as a whole —
// the reference skips an iteration whose loss is NaN (train.py:225-226); with data-parallel ranks the all-reduced bucket is
// non-finite on EVERY rank as soon as one rank's loss was, so all ranks take the same decision without a host round trip.
__global__ __launch_bounds__(256) void grad_finite_kernel(const float* __restrict__ g, long n, int* __restrict__ flag)
{
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = g[i];
        bad |= !(fabsf(v) <= 3.0e38f);                      // false for NaN and Inf
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
}

__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ buf,
                                                   long n, float lr, float momentum, float wd, float grad_scale, int first, int* __restrict__ flag)
{
    if (flag && flag[0]) {                                  // non-finite gradient: leave parameters and momentum untouched
        if (blockIdx.x == 0 && threadIdx.x == 0) atomicAdd(&flag[1], 1);
        return;
    }
    const long n4 = n >> 2;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 pv = reinterpret_cast<float4*>(p)[i];
        const float4 gv = reinterpret_cast<const float4*>(g)[i];
        float4 bv = first ? make_float4(0.f, 0.f, 0.f, 0.f) : reinterpret_cast<float4*>(buf)[i];
        const float gx = gv.x * grad_scale + wd * pv.x, gy = gv.y * grad_scale + wd * pv.y;
        const float gz = gv.z * grad_scale + wd * pv.z, gw = gv.w * grad_scale + wd * pv.w;
        bv.x = first ? gx : momentum * bv.x + gx; bv.y = first ? gy : momentum * bv.y + gy;
        bv.z = first ? gz : momentum * bv.z + gz; bv.w = first ? gw : momentum * bv.w + gw;
        pv.x -= lr * bv.x; pv.y -= lr * bv.y; pv.z -= lr * bv.z; pv.w -= lr * bv.w;
        reinterpret_cast<float4*>(buf)[i] = bv;
        reinterpret_cast<float4*>(p)[i] = pv;
    }
    const long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x;        // tail (n % 4 elements)
    if (i < n) {
        const float gg = g[i] * grad_scale + wd * p[i];
        const float b = first ? gg : momentum * buf[i] + gg;
        buf[i] = b;
        p[i] -= lr * b;
    }
}

// ModelEMA.update (utils/misc.py:76-86): v = v * d + (1 - d) * m, with torch's rounding sequence (three separately rounded
// float32 operations, d and 1-d rounded to float32 first) so that the result is bit-identical to the reference's.
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ v, const float* __restrict__ m, long n, float d, float omd)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
        v[i] = __fadd_rn(__fmul_rn(v[i], d), __fmul_rn(omd, m[i]));
}

void launch_ema(float* v, const float* m, long n, float d, float omd, hipStream_t s)
{
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(ema_kernel, dim3((unsigned)blocks), dim3(256), 0, s, v, m, n, d, omd);
}

void launch_sgd(float* p, const float* g, float* buf, long n, float lr, float momentum, float wd, float grad_scale, int first, int* flag, hipStream_t s)
{
    long blocks = ((n >> 2) + 255) / 256;
    if (blocks > 256 * 8) blocks = 256 * 8;
    if (blocks < 1) blocks = 1;
    if (flag) {
        (void)hipMemsetAsync(flag, 0, sizeof(int), s);
        hipLaunchKernelGGL(grad_finite_kernel, dim3((unsigned)(blocks > 512 ? 512 : blocks)), dim3(256), 0, s, g, n, flag);
    }
    hipLaunchKernelGGL(sgd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, buf, n, lr, momentum, wd, grad_scale, first, flag);
}

}  // namespace ynk

// kernels_post.hip — score head, box decode and per-class NMS on gfx950.
//
// Compiled with -ffp-contract=off: the NMS arithmetic must be the same sequence of IEEE binary32
// operations numpy performs in models/yolo_nano.py:159-188 (kept-index sets are compared bit-exactly).
//
//   decode_cand_kernel   models/yolo_nano.py:308-330 (head split) + :120-156 (decode) + :365-367 (scores)
//                        + :253-261 (argmax, threshold) fused: raw NHWC heads -> (box, best score, class)
//   score_full_kernel    same front end, writing the reference's all_bbox [N,4] / all_class [N,C]
//   argmax_cand_kernel   :253-261 from caller-provided (all_local, all_conf)
//   bucket_kernel        groups the surviving candidates of one image by class (LDS histogram + scatter)
//   nms_kernel           one wavefront per (image, class): greedy NMS by repeated wave-wide arg-max —
//                        no sort; pick order == descending score, equal scores: higher index first
//   compact_kernel       kept candidates in ascending candidate order (:274-277)
#include "yn_internal.h"

namespace ynk {

__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }

__device__ __forceinline__ void cand_location(const GridInfo& g, int n, int& s, int& cell, int& a)
{
    s = (n >= g.off[2]) ? 2 : ((n >= g.off[1]) ? 1 : 0);
    const int local = n - g.off[s];
    cell = local / g.A;
    a = local - cell * g.A;
}

__device__ __forceinline__ void decode_one(const GridInfo& g, int s, int cell, int a, const float* t, float S, float* box, bool normalise)
{
    const int gy = cell / g.w[s], gx = cell - gy * g.w[s];
    const float stride = (float)(8 << s);
    const float cx = (sigmoid_f(t[0]) + (float)gx) * stride;
    const float cy = (sigmoid_f(t[1]) + (float)gy) * stride;
    const float bw = expf(t[2]) * g.anchors[(s * g.A + a) * 2 + 0];
    const float bh = expf(t[3]) * g.anchors[(s * g.A + a) * 2 + 1];
    box[0] = cx - bw / 2; box[1] = cy - bh / 2; box[2] = cx + bw / 2; box[3] = cy + bh / 2;
    if (normalise) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = box[k] / S;
            box[k] = fminf(fmaxf(v, 0.0f), 1.0f);
        }
    }
}

template <bool FULL>
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ h0, const float* __restrict__ h1, const float* __restrict__ h2,
                                                      GridInfo g, int B, float conf_thresh,
                                                      float* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ cls,
                                                      float* __restrict__ all_class)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * g.N) return;
    const int b = (int)(i / g.N), n = (int)(i - (long)b * g.N);
    int s, cell, a;
    cand_location(g, n, s, cell, a);
    const float* head = s == 0 ? h0 : (s == 1 ? h1 : h2);
    const int HC = g.A * (5 + g.C);
    const float* row = head + ((size_t)b * g.hw[s] + cell) * HC;
    const float obj = sigmoid_f(row[a]);
    const float* cl = row + g.A + a * g.C;
    float mx = -INFINITY;
    for (int c = 0; c < g.C; ++c) mx = fmaxf(mx, cl[c]);
    float sum = 0.0f;
    for (int c = 0; c < g.C; ++c) sum += expf(cl[c] - mx);
    float best = -INFINITY;
    int bi = 0;
    for (int c = 0; c < g.C; ++c) {
        const float p = expf(cl[c] - mx) / sum * obj;
        if (FULL) all_class[(size_t)i * g.C + c] = p;
        if (p > best) { best = p; bi = c; }
    }
    float box[4];
    decode_one(g, s, cell, a, row + g.A * (1 + g.C) + a * 4, (float)g.S, box, true);
    *reinterpret_cast<float4*>(boxes + (size_t)i * 4) = make_float4(box[0], box[1], box[2], box[3]);
    if (!FULL) {
        scores[i] = best;
        cls[i] = (best >= conf_thresh) ? bi : -1;
    }
}

void launch_score_full(const float* const heads[3], const GridInfo& g, int B, float* all_bbox, float* all_class, hipStream_t s)
{
    const long total = (long)B * g.N;
    hipLaunchKernelGGL(decode_kernel<true>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       heads[0], heads[1], heads[2], g, B, 0.0f, all_bbox, (float*)nullptr, (int32_t*)nullptr, all_class);
}

void launch_decode_cand(const float* const heads[3], const GridInfo& g, int B, float conf_thresh,
                        float* boxes, float* scores, int32_t* cls, hipStream_t s)
{
    const long total = (long)B * g.N;
    hipLaunchKernelGGL(decode_kernel<false>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       heads[0], heads[1], heads[2], g, B, conf_thresh, boxes, scores, cls, (float*)nullptr);
}

// YOLONano.decode_boxes: txtytwth [B, sumHW, A, 4] -> xyxy pixels [B, N, 4]
__global__ __launch_bounds__(256) void decode_boxes_kernel(const float* __restrict__ t, GridInfo g, int B, float* __restrict__ out)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long)B * g.N) return;
    const int n = (int)(i % g.N);
    int s, cell, a;
    cand_location(g, n, s, cell, a);
    float box[4];
    decode_one(g, s, cell, a, t + (size_t)i * 4, (float)g.S, box, false);
    *reinterpret_cast<float4*>(out + (size_t)i * 4) = make_float4(box[0], box[1], box[2], box[3]);
}

void launch_decode_boxes(const float* txtytwth, const GridInfo& g, int B, float* xyxy, hipStream_t s)
{
    const long total = (long)B * g.N;
    hipLaunchKernelGGL(decode_boxes_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, txtytwth, g, B, xyxy);
}

// np.argmax (first maximum) + score gather + threshold — models/yolo_nano.py:253-261
__global__ __launch_bounds__(256) void argmax_cand_kernel(const float* __restrict__ all_local, const float* __restrict__ all_conf,
                                                           long total, int C, float conf_thresh,
                                                           float* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ cls)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const float* p = all_conf + (size_t)i * C;
    float best = p[0];
    int bi = 0;
    for (int c = 1; c < C; ++c) { const float v = p[c]; if (v > best) { best = v; bi = c; } }
    scores[i] = best;
    cls[i] = (best >= conf_thresh) ? bi : -1;
    if (boxes != all_local) *reinterpret_cast<float4*>(boxes + (size_t)i * 4) = *reinterpret_cast<const float4*>(all_local + (size_t)i * 4);
}

void launch_argmax_cand(const float* all_local, const float* all_conf, int B, int N, int C, float conf_thresh,
                        float* boxes, float* scores, int32_t* cls, hipStream_t s)
{
    const long total = (long)B * N;
    hipLaunchKernelGGL(argmax_cand_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s,
                       all_local, all_conf, total, C, conf_thresh, boxes, scores, cls);
}

// -------------------------------------------------------------------------------------------------
// Bucket the valid candidates of image b by class.  One workgroup per image; histogram and cursors in LDS.
// -------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void bucket_kernel(const int32_t* __restrict__ cls, int N, int C,
                                                       int32_t* __restrict__ seg_count, int32_t* __restrict__ seg_off,
                                                       int32_t* __restrict__ bucket, int32_t* __restrict__ keep)
{
    extern __shared__ int32_t lds[];            // hist[C], cursor[C]
    int32_t* hist = lds;
    int32_t* cursor = lds + C;
    const int b = blockIdx.x;
    const int32_t* c_in = cls + (size_t)b * N;
    for (int c = threadIdx.x; c < C; c += blockDim.x) hist[c] = 0;
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        keep[(size_t)b * N + n] = 0;
        const int c = c_in[n];
        if (c >= 0) atomicAdd(&hist[c], 1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int run = 0;
        for (int c = 0; c < C; ++c) {
            const int h = hist[c];
            seg_count[(size_t)b * C + c] = h;
            seg_off[(size_t)b * C + c] = run;
            cursor[c] = run;
            run += h;
        }
    }
    __syncthreads();
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        const int c = c_in[n];
        if (c >= 0) {
            const int pos = atomicAdd(&cursor[c], 1);
            bucket[(size_t)b * N + pos] = n;
        }
    }
}

// -------------------------------------------------------------------------------------------------
// Greedy NMS for one segment, executed by ONE wavefront.
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned order_bits(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // unsigned order == float order
}

__device__ __forceinline__ unsigned long long wave_max_u64(unsigned long long v)
{
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const unsigned lo = __shfl_xor((unsigned)(v & 0xffffffffu), off);
        const unsigned hi = __shfl_xor((unsigned)(v >> 32), off);
        const unsigned long long o = ((unsigned long long)hi << 32) | lo;
        v = o > v ? o : v;
    }
    return v;
}

// true when box j must be REMOVED given kept box i  (reference keeps `ovr <= thresh`; NaN -> removed)
__device__ __forceinline__ bool suppressed(const float4 bi, float ai, const float4 bj, float aj, float thresh, int diou)
{
    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
    const float w = fmaxf(1e-28f, xx2 - xx1);
    const float h = fmaxf(1e-28f, yy2 - yy1);
    const float inter = w * h;
    const float t0 = ai + aj;
    float ovr = inter / (t0 - inter);
    if (diou) {                                            // models/yolo_nano.py:216-236
        const float mxx = fmaxf(fmaxf(bi.x, bi.z), fmaxf(bj.x, bj.z)), mnx = fminf(fminf(bi.x, bi.z), fminf(bj.x, bj.z));
        const float mxy = fmaxf(fmaxf(bi.y, bi.w), fmaxf(bj.y, bj.w)), mny = fminf(fminf(bi.y, bi.w), fminf(bj.y, bj.w));
        const float dx = mxx - mnx, dy = mxy - mny;
        const float Cd = sqrtf(dx * dx + dy * dy);
        const float p1x = (bi.x + bi.z) / 2.0f, p1y = (bi.y + bi.w) / 2.0f;
        const float p2x = (bj.x + bj.z) / 2.0f, p2y = (bj.y + bj.w) / 2.0f;
        const float ex = p2x - p1x, ey = p2y - p1y;
        const float D = sqrtf(ex * ex + ey * ey);
        const float lens = (D * D) / (Cd * Cd + 1e-20f);
        ovr = ovr - lens;
    }
    return !(ovr <= thresh);
}

// ids == nullptr: the segment is items 0..n-1 of (boxes, scores) themselves (yn_nms single-class entry)
template <int T>
__device__ void nms_segment_regs(const float* __restrict__ boxes, const float* __restrict__ scores, const int32_t* __restrict__ ids,
                                 int n, float thresh, int diou, int32_t* __restrict__ keep_flags,
                                 int32_t* __restrict__ pick_list, int32_t* __restrict__ pick_count, float* sh)
{
    const int lane = threadIdx.x & 63;
    float4 bx[T];
    float ar[T];
    unsigned long long key[T];
    int id[T];
    bool alive[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int j = lane + 64 * t;
        alive[t] = j < n;
        id[t] = 0; key[t] = 0; ar[t] = 0.0f; bx[t] = make_float4(0, 0, 0, 0);
        if (alive[t]) {
            id[t] = ids ? ids[j] : j;
            bx[t] = *reinterpret_cast<const float4*>(boxes + (size_t)id[t] * 4);
            ar[t] = (bx[t].z - bx[t].x) * (bx[t].w - bx[t].y);
            key[t] = ((unsigned long long)order_bits(scores[id[t]]) << 32) | (unsigned)id[t];
        }
    }
    int picked = 0;
    while (true) {
        unsigned long long m = 0;
#pragma unroll
        for (int t = 0; t < T; ++t) if (alive[t] && key[t] > m) m = key[t];
        m = wave_max_u64(m);
        if (m == 0) break;
#pragma unroll
        for (int t = 0; t < T; ++t) {
            if (alive[t] && key[t] == m) {
                sh[0] = bx[t].x; sh[1] = bx[t].y; sh[2] = bx[t].z; sh[3] = bx[t].w; sh[4] = ar[t];
                alive[t] = false;
                if (keep_flags) keep_flags[id[t]] = 1;
                if (pick_list) pick_list[picked] = id[t];
            }
        }
        ++picked;
        __syncthreads();
        const float4 bi = make_float4(sh[0], sh[1], sh[2], sh[3]);
        const float ai = sh[4];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < T; ++t)
            if (alive[t] && suppressed(bi, ai, bx[t], ar[t], thresh, diou)) alive[t] = false;
    }
    if (pick_count && lane == 0) *pick_count = picked;
}

// arbitrary n: alive flags live in `state` (global scratch, n ints), data re-read each round (L2-resident)
__device__ void nms_segment_mem(const float* __restrict__ boxes, const float* __restrict__ scores, const int32_t* __restrict__ ids,
                                int n, float thresh, int diou, int32_t* __restrict__ keep_flags,
                                int32_t* __restrict__ pick_list, int32_t* __restrict__ pick_count,
                                int32_t* __restrict__ state, float* sh)
{
    const int lane = threadIdx.x & 63;
    for (int j = lane; j < n; j += 64) state[j] = 1;
    int picked = 0;
    while (true) {
        unsigned long long m = 0;
        for (int j = lane; j < n; j += 64) {
            if (state[j]) {
                const int idj = ids ? ids[j] : j;
                const unsigned long long k = ((unsigned long long)order_bits(scores[idj]) << 32) | (unsigned)idj;
                if (k > m) m = k;
            }
        }
        m = wave_max_u64(m);
        if (m == 0) break;
        const int win = (int)(unsigned)(m & 0xffffffffu);
        const float4 bi = *reinterpret_cast<const float4*>(boxes + (size_t)win * 4);
        const float ai = (bi.z - bi.x) * (bi.w - bi.y);
        if (lane == 0) {
            if (keep_flags) keep_flags[win] = 1;
            if (pick_list) pick_list[picked] = win;
        }
        ++picked;
        for (int j = lane; j < n; j += 64) {
            if (state[j]) {
                const int idj = ids ? ids[j] : j;
                if (idj == win) { state[j] = 0; continue; }
                const float4 bj = *reinterpret_cast<const float4*>(boxes + (size_t)idj * 4);
                const float aj = (bj.z - bj.x) * (bj.w - bj.y);
                if (suppressed(bi, ai, bj, aj, thresh, diou)) state[j] = 0;
            }
        }
    }
    if (pick_count && lane == 0) *pick_count = picked;
    (void)sh;
}

__device__ void nms_segment(const float* boxes, const float* scores, const int32_t* ids, int n, float thresh, int diou,
                            int32_t* keep_flags, int32_t* pick_list, int32_t* pick_count, int32_t* state, float* sh)
{
    if (n <= 64) nms_segment_regs<1>(boxes, scores, ids, n, thresh, diou, keep_flags, pick_list, pick_count, sh);
    else if (n <= 256) nms_segment_regs<4>(boxes, scores, ids, n, thresh, diou, keep_flags, pick_list, pick_count, sh);
    else nms_segment_mem(boxes, scores, ids, n, thresh, diou, keep_flags, pick_list, pick_count, state, sh);
}

__global__ __launch_bounds__(64) void nms_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                  const int32_t* __restrict__ seg_count, const int32_t* __restrict__ seg_off,
                                                  const int32_t* __restrict__ bucket, int N, int C, float thresh, int diou,
                                                  int32_t* __restrict__ keep, int32_t* __restrict__ state)
{
    __shared__ float sh[8];
    const int c = blockIdx.x, b = blockIdx.y;
    const int n = seg_count[(size_t)b * C + c];
    if (n == 0) return;
    const int off = seg_off[(size_t)b * C + c];
    nms_segment(boxes + (size_t)b * N * 4, scores + (size_t)b * N, bucket + (size_t)b * N + off, n, thresh, diou,
                keep + (size_t)b * N, nullptr, nullptr, state + (size_t)b * N + off, sh);
}

__global__ __launch_bounds__(64) void nms_single_kernel(const float* __restrict__ dets, const float* __restrict__ scores, int n,
                                                         float thresh, int diou, int32_t* __restrict__ state,
                                                         int32_t* __restrict__ keep, int32_t* __restrict__ count)
{
    __shared__ float sh[8];
    if (n <= 0) { if (threadIdx.x == 0) *count = 0; return; }
    nms_segment(dets, scores, nullptr, n, thresh, diou, nullptr, keep, count, state, sh);
}

// kept candidates of image b, ascending candidate index (np.where(keep > 0), models/yolo_nano.py:274-277)
__global__ __launch_bounds__(1024) void compact_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                        const int32_t* __restrict__ cls, const int32_t* __restrict__ keep, int N,
                                                        float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                        int32_t* __restrict__ out_cls, int32_t* __restrict__ out_index,
                                                        int32_t* __restrict__ count)
{
    __shared__ int wave_sums[16];
    __shared__ int base;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int n0 = 0; n0 < N; n0 += 1024) {
        const int n = n0 + threadIdx.x;
        const int f = (n < N && keep[(size_t)b * N + n]) ? 1 : 0;
        const unsigned long long bal = __ballot(f);
        const int before = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_sums[wave] = __popcll(bal);
        __syncthreads();
        int wbase = 0, tot = 0;
        for (int w = 0; w < 16; ++w) { const int v = wave_sums[w]; if (w < wave) wbase += v; tot += v; }
        const int pos = base + wbase + before;
        if (f) {
            const size_t src = (size_t)b * N + n, dst = (size_t)b * N + pos;
            *reinterpret_cast<float4*>(out_boxes + dst * 4) = *reinterpret_cast<const float4*>(boxes + src * 4);
            out_scores[dst] = scores[src];
            out_cls[dst] = cls[src];
            if (out_index) out_index[dst] = n;
        }
        __syncthreads();
        if (threadIdx.x == 0) base += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) count[b] = base;
}

void launch_nms_pipeline(const float* boxes, const float* scores, const int32_t* cls, int B, int N, int C,
                         float nms_thresh, int diou, const NmsWork& wk,
                         float* out_boxes, float* out_scores, int32_t* out_cls, int32_t* out_index, int32_t* count,
                         hipStream_t s)
{
    hipLaunchKernelGGL(bucket_kernel, dim3(B), dim3(1024), 2 * C * sizeof(int32_t), s, cls, N, C, wk.seg_count, wk.seg_off, wk.bucket, wk.keep);
    hipLaunchKernelGGL(nms_kernel, dim3(C, B), dim3(64), 0, s, boxes, scores, wk.seg_count, wk.seg_off, wk.bucket, N, C, nms_thresh, diou, wk.keep, wk.state);
    hipLaunchKernelGGL(compact_kernel, dim3(B), dim3(1024), 0, s, boxes, scores, cls, wk.keep, N, out_boxes, out_scores, out_cls, out_index, count);
}

void launch_nms_single(const float* dets, const float* scores, int n, float thresh, int diou,
                       int32_t* state_scratch, int32_t* keep, int32_t* count, hipStream_t s)
{
    hipLaunchKernelGGL(nms_single_kernel, dim3(1), dim3(64), 0, s, dets, scores, n, thresh, diou, state_scratch, keep, count);
}

}  // namespace ynk

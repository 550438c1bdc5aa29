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

// -------------------------------------------------------------------------------------------------
// Greedy NMS for one segment by ONE workgroup (256 threads): sort + 64-wide chunk resolve.
//   1. keys (score bits << 32 | id) are bitonic-sorted in LDS, descending  => pick order.
//   2. chunk c = sorted items [64c, 64c+64): wave 0 builds the 64x64 suppression bit-matrix of the
//      chunk (lane i vs lanes t > i), then resolves it serially with scalar readlanes — exactly the
//      reference's while-loop restricted to the chunk;
//   3. every thread tests the not-yet-removed items behind the chunk against the chunk's kept boxes.
// The serial depth is n/64 chunk rounds instead of one round per kept box.
// LDS carve (dynamic): keys[P] u64 | removed[P] u8 | cbox[64] float4 | carea[64] | kbox[64] float4 | karea[64] | misc
// -------------------------------------------------------------------------------------------------
__device__ void nms_sorted_block(const float* __restrict__ boxes, const float* __restrict__ scores, const int32_t* __restrict__ ids,
                                 int n, int P, float thresh, int diou, int32_t* __restrict__ keep_flags,
                                 int32_t* __restrict__ pick_list, int32_t* __restrict__ pick_count, unsigned char* lds)
{
    unsigned long long* keys = reinterpret_cast<unsigned long long*>(lds);
    unsigned char* removed = lds + (size_t)P * 8;
    float4* cbox = reinterpret_cast<float4*>(lds + (size_t)P * 9);
    float* carea = reinterpret_cast<float*>(cbox + 64);
    float4* kbox = reinterpret_cast<float4*>(carea + 64);
    float* karea = reinterpret_cast<float*>(kbox + 64);
    int* misc = reinterpret_cast<int*>(karea + 64);          // [0] = kept in this chunk
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    for (int j = tid; j < P; j += 256) {
        unsigned long long k = 0;                             // padding sorts to the end
        if (j < n) {
            const int id = ids ? ids[j] : j;
            k = ((unsigned long long)order_bits(scores[id]) << 32) | (unsigned)id;
        }
        keys[j] = k;
        removed[j] = 0;
    }
    __syncthreads();
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (P >> 1); i += 256) {
                const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int b = a + j;
                const unsigned long long x = keys[a], y = keys[b];
                const bool desc = (a & k) == 0;
                if ((x < y) == desc) { keys[a] = y; keys[b] = x; }
            }
            __syncthreads();
        }
    }
    int picked = 0;
    const int nchunks = (n + 63) >> 6;
    for (int c = 0; c < nchunks; ++c) {
        const int base = c << 6;
        if (wave == 0) {
            const int j = base + lane;
            const bool in = j < n;
            const bool valid = in && !removed[j];
            const int id = in ? (int)(unsigned)(keys[j] & 0xffffffffu) : 0;
            float4 bx = make_float4(0.f, 0.f, 0.f, 0.f);
            if (in) bx = *reinterpret_cast<const float4*>(boxes + (size_t)id * 4);
            const float ar = (bx.z - bx.x) * (bx.w - bx.y);
            cbox[lane] = bx;
            carea[lane] = ar;
            __builtin_amdgcn_wave_barrier();
            const unsigned long long alive0 = __ballot(valid);
            unsigned long long mask = 0;
            if (valid) {
                for (int t = lane + 1; t < 64; ++t) {
                    if (!((alive0 >> t) & 1ull)) continue;
                    if (suppressed(bx, ar, cbox[t], carea[t], thresh, diou)) mask |= 1ull << t;
                }
            }
            const unsigned mlo = (unsigned)(mask & 0xffffffffu), mhi = (unsigned)(mask >> 32);
            unsigned long long alive = alive0, keepm = 0;
            for (int i = 0; i < 64; ++i) {
                if ((alive >> i) & 1ull) {
                    keepm |= 1ull << i;
                    const unsigned lo_i = (unsigned)__builtin_amdgcn_readlane((int)mlo, i);     // (unsigned): no sign extension
                    const unsigned hi_i = (unsigned)__builtin_amdgcn_readlane((int)mhi, i);
                    const unsigned long long mi = ((unsigned long long)hi_i << 32) | (unsigned long long)lo_i;
                    alive &= ~mi;
                }
            }
            const int rank = __popcll(keepm & ((1ull << lane) - 1ull));
            if ((keepm >> lane) & 1ull) {
                kbox[rank] = bx;
                karea[rank] = ar;
                if (keep_flags) keep_flags[id] = 1;
                if (pick_list) pick_list[picked + rank] = id;
            }
            if (lane == 0) misc[0] = __popcll(keepm);
        }
        __syncthreads();
        const int nk = misc[0];
        picked += nk;
        if (nk > 0) {
            for (int j = base + 64 + tid; j < n; j += 256) {
                if (removed[j]) continue;
                const int id = (int)(unsigned)(keys[j] & 0xffffffffu);
                const float4 bj = *reinterpret_cast<const float4*>(boxes + (size_t)id * 4);
                const float aj = (bj.z - bj.x) * (bj.w - bj.y);
                for (int k = 0; k < nk; ++k)
                    if (suppressed(kbox[k], karea[k], bj, aj, thresh, diou)) { removed[j] = 1; break; }
            }
        }
        __syncthreads();
    }
    if (pick_count && tid == 0) *pick_count = picked;
}

__host__ __device__ inline int nms_pow2(int n) { int p = 64; while (p < n) p <<= 1; return p; }
__host__ __device__ inline size_t nms_lds_bytes(int P) { return (size_t)P * 9 + 64 * 16 * 2 + 64 * 4 * 2 + 64; }

#define YN_NMS_SMALL 1024
#define YN_NMS_LARGE 8192

// grid (C, B); handles the segments with n_lo < n <= n_hi (LDS sized for n_hi)
__global__ __launch_bounds__(256) void nms_sorted_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                          const int32_t* __restrict__ seg_count, const int32_t* __restrict__ seg_off,
                                                          const int32_t* __restrict__ bucket, int N, int C, float thresh, int diou,
                                                          int32_t* __restrict__ keep, int n_lo, int n_hi)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_lds[];
    const int c = blockIdx.x, b = blockIdx.y;
    const int n = seg_count[(size_t)b * C + c];
    if (n <= n_lo || n > n_hi) return;
    const int off = seg_off[(size_t)b * C + c];
    nms_sorted_block(boxes + (size_t)b * N * 4, scores + (size_t)b * N, bucket + (size_t)b * N + off, n, nms_pow2(n), thresh, diou,
                     keep + (size_t)b * N, nullptr, nullptr, nms_lds);
}

// segments too large for LDS (n > YN_NMS_LARGE): one wavefront, state in global scratch (correct, slow, rare)
__global__ __launch_bounds__(64) void nms_huge_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                       const int32_t* __restrict__ seg_count, const int32_t* __restrict__ seg_off,
                                                       const int32_t* __restrict__ bucket, int N, int C, float thresh, int diou,
                                                       int32_t* __restrict__ keep, int32_t* __restrict__ state)
{
    __shared__ float sh[8];
    const int c = blockIdx.x, b = blockIdx.y;
    const int n = seg_count[(size_t)b * C + c];
    if (n <= YN_NMS_LARGE) return;
    const int off = seg_off[(size_t)b * C + c];
    nms_segment_mem(boxes + (size_t)b * N * 4, scores + (size_t)b * N, bucket + (size_t)b * N + off, n, thresh, diou,
                    keep + (size_t)b * N, nullptr, nullptr, state + (size_t)b * N + off, sh);
}

__global__ __launch_bounds__(256) void nms_single_sorted_kernel(const float* __restrict__ dets, const float* __restrict__ scores, int n,
                                                                 float thresh, int diou, int32_t* __restrict__ keep, int32_t* __restrict__ count)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char nms_lds[];
    if (n <= 0) { if (threadIdx.x == 0) *count = 0; return; }
    nms_sorted_block(dets, scores, nullptr, n, nms_pow2(n), thresh, diou, nullptr, keep, count, nms_lds);
}

__global__ __launch_bounds__(64) void nms_single_huge_kernel(const float* __restrict__ dets, const float* __restrict__ scores, int n,
                                                              float thresh, int diou, int32_t* __restrict__ state,
                                                              int32_t* __restrict__ keep, int32_t* __restrict__ count)
{
    __shared__ float sh[8];
    nms_segment_mem(dets, scores, nullptr, n, thresh, diou, nullptr, keep, count, state, sh);
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
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sorted_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nms_lds_bytes(YN_NMS_LARGE));
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_single_sorted_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nms_lds_bytes(YN_NMS_LARGE));
        attr_set = true;
    }
    hipLaunchKernelGGL(bucket_kernel, dim3(B), dim3(1024), 2 * C * sizeof(int32_t), s, cls, N, C, wk.seg_count, wk.seg_off, wk.bucket, wk.keep);
    hipLaunchKernelGGL(nms_sorted_kernel, dim3(C, B), dim3(256), nms_lds_bytes(YN_NMS_SMALL), s, boxes, scores, wk.seg_count, wk.seg_off, wk.bucket,
                       N, C, nms_thresh, diou, wk.keep, 0, YN_NMS_SMALL);
    if (N > YN_NMS_SMALL)
        hipLaunchKernelGGL(nms_sorted_kernel, dim3(C, B), dim3(256), nms_lds_bytes(YN_NMS_LARGE), s, boxes, scores, wk.seg_count, wk.seg_off, wk.bucket,
                           N, C, nms_thresh, diou, wk.keep, YN_NMS_SMALL, YN_NMS_LARGE);
    if (N > YN_NMS_LARGE)
        hipLaunchKernelGGL(nms_huge_kernel, dim3(C, B), dim3(64), 0, s, boxes, scores, wk.seg_count, wk.seg_off, wk.bucket, N, C, nms_thresh, diou,
                           wk.keep, wk.state);
    hipLaunchKernelGGL(compact_kernel, dim3(B), dim3(1024), 0, s, boxes, scores, cls, wk.keep, N, out_boxes, out_scores, out_cls, out_index, count);
}

void launch_nms_single(const float* dets, const float* scores, int n, float thresh, int diou,
                       int32_t* state_scratch, int32_t* keep, int32_t* count, hipStream_t s)
{
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_single_sorted_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)nms_lds_bytes(YN_NMS_LARGE));
        attr_set = true;
    }
    if (n <= YN_NMS_LARGE)
        hipLaunchKernelGGL(nms_single_sorted_kernel, dim3(1), dim3(256), nms_lds_bytes(nms_pow2(n > 0 ? n : 1)), s, dets, scores, n, thresh, diou, keep, count);
    else
        hipLaunchKernelGGL(nms_single_huge_kernel, dim3(1), dim3(64), 0, s, dets, scores, n, thresh, diou, state_scratch, keep, count);
}

}  // namespace ynk

// kernels_post.hip — score head, box decode and per-class NMS on gfx950.
//
// Compiled with -ffp-contract=off: the NMS arithmetic must be the same sequence of IEEE binary32
// operations numpy performs in models/yolo_nano.py:159-188 (kept-index sets are compared bit-exactly).
//
//   decode_cand_kernel   models/yolo_nano.py:308-330 (head split) + :120-156 (decode) + :365-367 (scores)
//                        + :253-261 (argmax, threshold) fused: raw NHWC heads -> (box, best score, class)
//   score_full_kernel    same front end, writing the reference's all_bbox [N,4] / all_class [N,C]
//   argmax_cand_kernel   :253-261 from caller-provided (all_local, all_conf)
//   bucket / sort / matrix / resolve / compact : exact per-class greedy NMS (see the block comment below)
#include "yn_device.h"

namespace ynk {

__device__ __forceinline__ float sigmoid_f(float v) { return 1.0f / (1.0f + expf(-v)); }
// exp(x) for the class softmax's x = logit - max <= 0: one multiply + v_exp_f32 instead of expf's ~15 instructions (range reduction, ldexp, overflow /
// denormal selects - none of which x <= 0 needs: a result below 2^-126 adds nothing to a sum >= 1).  80 of these per candidate were a third of
// head_tail_group_kernel's class pass, and that kernel is VALU-issue bound.  exp_le0(0) == 1 exactly (the arg-max rule relies on it); the
// error against expf is <= 2 ulp near 0 and <= |x| * 1e-7 relative for the far classes - 1e-6 on a score, two orders inside the parity tolerance (1e-4).
__device__ __forceinline__ float exp_le0(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

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

__device__ __forceinline__ unsigned f32_order_bits(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

// Reductions over a 16-lane DPP row.  Lane i combines with lane (i + r) % 16 for r = 8, 4, 2, 1 (row_ror): after each step
// lanes i and i ^ r hold equal values, so this is the xor butterfly — the same combination tree, hence the same fp32 sum bit
// for bit — as one DPP-modified VALU instruction per step; __shfl_xor compiles to ds_bpermute_b32, whose ~100-cycle LDS round
// trip per step (14 dependent steps per candidate) made the decode latency-bound at 4 waves per SIMD.
template <int CTRL> __device__ __forceinline__ float dpp_f(float v)
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL> __device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
#define YN_ROW_STEPS(OP) OP(0x128) OP(0x124) OP(0x122) OP(0x121)
__device__ __forceinline__ float group16_max(float v)
{
#define OP(c) v = fmaxf(v, dpp_f<c>(v));
    YN_ROW_STEPS(OP)
#undef OP
    return v;
}
__device__ __forceinline__ float group16_sum(float v)
{
#define OP(c) v += dpp_f<c>(v);
    YN_ROW_STEPS(OP)
#undef OP
    return v;
}
__device__ __forceinline__ unsigned long long group16_max_u64(unsigned long long v)
{
#define OP(c) { const unsigned lo = (unsigned)dpp_i<c>((int)(unsigned)(v & 0xffffffffu)), hi = (unsigned)dpp_i<c>((int)(unsigned)(v >> 32));          \
                const unsigned long long o = ((unsigned long long)hi << 32) | lo; v = o > v ? o : v; }
    YN_ROW_STEPS(OP)
#undef OP
    return v;
}
__device__ __forceinline__ int group16_min_i(int v)
{
#define OP(c) { const int o = dpp_i<c>(v); v = o < v ? o : v; }
    YN_ROW_STEPS(OP)
#undef OP
    return v;
}

// one coordinate (k = 0..3: x1, y1, x2, y2) of decode_one's normalised box: the same operations in the same order, so the four
// lanes that each evaluate one coordinate reproduce decode_one bit for bit (tc = t[k & 1], ts = t[2 + (k & 1)])
__device__ __forceinline__ float decode_coord(const GridInfo& g, int s, int gx, int gy, int a, float tc, float ts, float S, int k)
{
    const float stride = (float)(8 << s);
    const float c = (sigmoid_f(tc) + (float)((k & 1) ? gy : gx)) * stride;
    const float e = expf(ts) * g.anchors[(s * g.A + a) * 2 + (k & 1)];
    const float v = ((k < 2) ? c - e / 2 : c + e / 2) / S;
    return fminf(fmaxf(v, 0.0f), 1.0f);
}

// One candidate on 16 lanes (j = lane in the group): softmax over the classes x sigmoid(objectness), arg-max with numpy's
// first-maximum rule, box decode (models/yolo_nano.py:253-261, 308-330, 362-367).  row = the pixel's raw head values
// [A obj | A*C classes | A*4 box]; (gx, gy) = the cell.  The score of class c is p_c = e_c / sum * obj with e_c = exp(x_c - max):
// p is monotone in e and the maximal e is exactly 1, so unless another class sits within 1e-5 of the maximum (or the product
// underflows) the winner is the first class with e == 1 and its score 1 / sum * obj — ONE division per candidate; any wavefront
// holding a near-tie takes the general path that evaluates every p_c (identical results, pinned by the parity suite).
// Class part of one candidate on its 16 lanes (4 candidates per wavefront): lane j of a group owns classes j, j+16, ...; the softmax
// max / sum / arg-max are 4-step reductions inside the 16-lane DPP row.  KMAX*16 >= C.  Returns true when this WAVEFRONT took the general path — then `sc` / `cbest` are the
// final score and class (needs the objectness) — else the caller finishes with score = 1 / sum * sigmoid(obj_raw), class = cbest.
// The general path also covers obj_raw < -60: sigmoid below 1e-26, where score = obj / sum (sum <= C) may underflow to equal products.
// (two halves, so that a caller can run the branch-free first half of SEVERAL candidates back to back - independent dependency chains
// the scheduler interleaves - before any of the wave-level decisions: head_tail_block's pass A)
template <bool FULL, int KMAX>
__device__ __forceinline__ void cand_class_stats(const GridInfo& g, const float* row, int a, int j, float obj_raw,
                                                 float (&v)[KMAX], float& sum_out, int& first_out, bool& general_out)
{
    const float* cl = row + g.A + a * g.C;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {                                      // clamped index, -inf through an opaque mask for the slots past C
        const int c = j + 16 * k;
        unsigned mk = c < g.C ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(mk));
        const unsigned bits = __float_as_uint(cl[c < g.C ? c : g.C - 1]);
        v[k] = __uint_as_float((bits & mk) | (0xff800000u & ~mk));       // -inf
    }
    float mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) mx = fmaxf(mx, v[k]);
    mx = group16_max(mx);
    float sum = 0.0f;
    bool general = FULL || !(obj_raw >= -60.0f);
    int first = 0x7fffffff;
#pragma unroll
    for (int k = KMAX - 1; k >= 0; --k) {
        const int c = j + 16 * k;
        v[k] = c < g.C ? exp_le0(v[k] - mx) : 0.0f;
        if (v[k] == 1.0f) first = c;                                      // descending k: ends at this lane's lowest such class
        general = general | ((v[k] < 1.0f) & (v[k] > 0.99999f));           // bitwise: the short-circuit form compiled to an exec-mask branch per class slot
    }
#pragma unroll
    for (int k = 0; k < KMAX; ++k) sum += v[k];
    sum_out = group16_sum(sum);
    first_out = first;
    general_out = general;
}

template <bool FULL, int KMAX>
__device__ __forceinline__ bool cand_class_finish(const GridInfo& g, int i, int j, float obj_raw, const float (&v)[KMAX], float sum, int first, bool general,
                                                  float& sc, int& cbest, float* __restrict__ all_class)
{
    if (__any(general)) {
        const float obj = sigmoid_f(obj_raw);
        unsigned long long best = 0;
#pragma unroll
        for (int k = 0; k < KMAX; ++k) {
            const int c = j + 16 * k;
            if (c < g.C) {
                const float p = v[k] / sum * obj;
                if (FULL) all_class[(size_t)i * g.C + c] = p;
                const unsigned long long key = ((unsigned long long)f32_order_bits(p) << 32) | (unsigned)(0x7fffffff - c);   // first max wins ties
                best = key > best ? key : best;
            }
        }
        if (!FULL) best = group16_max_u64(best);
        const unsigned ub = (unsigned)(best >> 32);
        sc = __uint_as_float((ub & 0x80000000u) ? (ub & 0x7fffffffu) : ~ub);
        cbest = (int)(0x7fffffff - (unsigned)(best & 0xffffffffu));
        return true;
    }
    cbest = group16_min_i(first);
    return false;
}

template <bool FULL, int KMAX>
__device__ __forceinline__ bool cand_class(const GridInfo& g, const float* row, int i, int a, int j, float obj_raw,
                                           float& sum_out, float& sc, int& cbest, float* __restrict__ all_class)
{
    float v[KMAX];
    int first;
    bool general;
    cand_class_stats<FULL, KMAX>(g, row, a, j, obj_raw, v, sum_out, first, general);
    return cand_class_finish<FULL, KMAX>(g, i, j, obj_raw, v, sum_out, first, general, sc, cbest, all_class);
}

// models/yolo_nano.py:253-261 keeps a candidate only when score >= conf_thresh, and score = p_class * sigmoid(obj) with p_class <= 1:
// in float arithmetic too, because e_c <= 1 <= sum and rounding is monotone, fl(fl(e_c / sum) * obj) <= obj.  So a candidate with
// sigmoid(obj) < conf_thresh is out whatever its classes are, and when that holds for all (four) candidates of a wavefront the class
// softmax - 80 exponentials, two row reductions, the arg-max - is skipped: score 0, class -1, exactly what the threshold would make of
// it.  The test runs in LOGIT space so that the hot path pays one compare, not an exponential and a division per candidate:
// obj_raw < logit(conf_thresh) - 0.01 implies sigmoid(obj_raw) < conf_thresh * (1 - ~1 %), a margin five orders of magnitude above the
// float error of either side; candidates inside the margin simply take the full path.  (A NaN objectness compares false; conf_thresh
// outside (0, 1) never skips.)  With a trained model at the usual deployment thresholds (0.1 ... 0.3) that is nearly every wavefront;
// with the benchmark's 0.001 on random weights none.
__device__ __forceinline__ float conf_skip_logit(float conf_thresh)
{
    return (conf_thresh > 0.0f && conf_thresh < 1.0f) ? logf(conf_thresh / (1.0f - conf_thresh)) - 0.01f : -INFINITY;
}
__device__ __forceinline__ bool wave_below_conf(float obj_raw, float skip_logit)
{
    return __all(obj_raw < skip_logit) != 0;
}

template <bool FULL, int KMAX>
__device__ __forceinline__ void decode_candidate(const GridInfo& g, const float* row, int i, int s, int gx, int gy, int a, int j, float conf_thresh,
                                                 float* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ cls,
                                                 float* __restrict__ all_class)
{
    // the candidate's loads are issued before any is used: objectness, one of the four box values, the class slice (cand_class)
    const float obj_raw = row[a];
    const float tbox = row[g.A * (1 + g.C) + a * 4 + (j & 3)];
    float sum, sc;
    int cbest;
    if (!FULL && wave_below_conf(obj_raw, conf_skip_logit(conf_thresh))) { sc = 0.0f; cbest = 0; }
    else if (!cand_class<FULL, KMAX>(g, row, i, a, j, obj_raw, sum, sc, cbest, all_class)) sc = 1.0f / sum * sigmoid_f(obj_raw);
    // lane j evaluates coordinate j & 3 from box values (j & 1) and 2 + (j & 1) of its row (row_newbcast: lane n of the row to all)
    const float tc = (j & 1) ? dpp_f<0x151>(tbox) : dpp_f<0x150>(tbox), ts = (j & 1) ? dpp_f<0x153>(tbox) : dpp_f<0x152>(tbox);
    const float coord = decode_coord(g, s, gx, gy, a, tc, ts, (float)g.S, j & 3);
    if (j < 4) boxes[(size_t)i * 4 + j] = coord;
    if (!FULL && j == 0) {
        scores[i] = sc;
        cls[i] = (sc >= conf_thresh) ? cbest : -1;
    }
}

template <bool FULL, int KMAX>
__global__ __launch_bounds__(256) void decode_kernel(const float* __restrict__ h0, const float* __restrict__ h1, const float* __restrict__ h2,
                                                      GridInfo g, int B, float conf_thresh,
                                                      float* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ cls,
                                                      float* __restrict__ all_class)
{
    const int j = threadIdx.x & 15;
    const int total = B * g.N;
    const int HC = g.head_ld;
    for (int i = (int)(((long)blockIdx.x * 256 + threadIdx.x) >> 4); i < total; i += (int)(((long)gridDim.x * 256) >> 4)) {
        const int b = i / g.N, n = i - b * g.N;
        int s, cell, a;
        cand_location(g, n, s, cell, a);                                  // candidate order of models/yolo_nano.py:308-330
        const float* head = s == 0 ? h0 : (s == 1 ? h1 : h2);
        const int gy = cell / g.w[s], gx = cell - gy * g.w[s];
        decode_candidate<FULL, KMAX>(g, head + ((size_t)b * g.hw[s] + cell) * HC, i, s, gx, gy, a, j, conf_thresh, boxes, scores, cls, all_class);
    }
}

// The last pointwise conv of a detection head (models/yolo_nano.py:299-301, K = 96 -> A(5+C) columns) and the candidate decode
// of its scale (:308-330, 362-367) in one kernel: the raw head tile of 32 pixels never leaves the CU.  GEMM = gemm_split_tile as
// one 32 x (128*NT) block (all columns of a pixel), bias added into an LDS tile [32][128*NT], then decode_candidate — the code
// the standalone decode_kernel runs, 16 lanes per candidate — reads its row from LDS instead of HBM: results are bit-identical to
// head GEMM + decode_kernel (the parity suite pins that), the 4*A(5+C) bytes per pixel of raw head are neither written nor re-read.
template <int NT> struct HeadDecodeLds {
    static constexpr int BM = 32, BN = 128 * NT, LD = BN + 4;
    static constexpr int GEMM_HALVES = gemm_split_smem_halves(BM, BN), RAW_HALVES = BM * LD * 2 + BM * 8 * 4 + BM * 4 * 2;     // raw tile + [BM * A <= BM * 8] (sum, class) + [BM] (first candidate, gx, gy)
    static constexpr int HALVES = GEMM_HALVES > RAW_HALVES ? GEMM_HALVES : RAW_HALVES;
};

template <int NT, int KMAX>
__device__ __forceinline__ void head_decode_block(const GemmArgs& a, const GridInfo& g, int scale, float conf_thresh,
                                                  float* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ cls, int dbg,
                                                  c3h16* smem, unsigned bid, unsigned nblocks)
{
    constexpr int BM = 32, BN = 128 * NT, LD = BN + 4;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int m0 = (int)(((bid & 7u) * (nblocks >> 3)) + (bid >> 3)) * BM;      // XCD x streams a contiguous run of rows
    if (m0 >= a.M) return;
    f32x16 acc[NT];
    if (dbg & 1) {
        for (int nt = 0; nt < NT; ++nt) for (int r = 0; r < 16; ++r) acc[nt][r] = 0.01f * (float)(r + lane);
    } else
    gemm_split_tile<1, 4, NT>(a, smem, m0, 0, acc);
    __syncthreads();                                        // every wave is past its last operand read: the raw tile reuses the space
    float* raw = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = wave * NT * 32 + nt * 32 + l31;
        const float bias = a.bias[n < a.Npad ? n : 0];
#pragma unroll
        for (int r = 0; r < 16; ++r) raw[((r & 3) + 8 * (r >> 2) + 4 * h) * LD + n] = acc[nt][r] + bias;
    }
    // per row of the tile: index of its first candidate and its cell — the divisions by the map size happen once per pixel here, not
    // once per candidate / coordinate in the passes below (three integer divisions were ~75 of their ~200 instructions per item)
    int* rowinfo = reinterpret_cast<int*>(raw + BM * LD + BM * 16);     // [BM][4]
    if (t < BM) {
        const int m = min(m0 + t, a.M - 1);
        const int b = m / g.hw[scale], cell = m - b * g.hw[scale];
        const int gy = cell / g.w[scale], gx = cell - gy * g.w[scale];
        *reinterpret_cast<int4*>(rowinfo + t * 4) = make_int4(b * g.N + g.off[scale] + cell * g.A, gx, gy, 0);
    }
    const unsigned magicA = (65536u + (unsigned)g.A - 1u) / (unsigned)g.A;     // c / A == (c * magicA) >> 16 for c < BM * A <= 256
    __syncthreads();
    if (dbg & 2) return;
    // Decode in three passes so that no per-candidate scalar work is replicated over 16 lanes:
    //   A  16 lanes per candidate, 16 candidates per pass: the class softmax statistics (cand_class) -> LDS
    //   B  one thread per candidate: objectness sigmoid, score, threshold -> scores / cls
    //   C  one thread per box coordinate (decode_coord) -> boxes
    // the arithmetic per value is decode_candidate's, so the outputs equal head GEMM + decode_kernel bit for bit.
    float* st_sum = reinterpret_cast<float*>(raw + BM * LD);              // [BM * A] sum of exponentials, or the final score (general path)
    int* st_cls = reinterpret_cast<int*>(st_sum + BM * g.A);              // [BM * A] class, bit 31 = st_sum holds the final score
    const int j = t & 15;
    const int ncand = min(BM, a.M - m0) * g.A;
    const float skip_logit = conf_skip_logit(conf_thresh);
    for (int c = t >> 4; c < ncand; c += 16) {
        const int row = (int)(((unsigned)c * magicA) >> 16), an = c - row * g.A;
        float sum, sc;
        int cbest;
        const float* rp = raw + row * LD;
        if (wave_below_conf(rp[an], skip_logit)) { if (j == 0) { st_sum[c] = 0.0f; st_cls[c] = (int)0x80000000; } continue; }
        const bool fin = cand_class<false, KMAX>(g, rp, 0, an, j, rp[an], sum, sc, cbest, nullptr);
        if (j == 0) { st_sum[c] = fin ? sc : sum; st_cls[c] = fin ? (cbest | (int)0x80000000) : cbest; }
    }
    __syncthreads();
    for (int q = t; q < 5 * ncand; q += 256) {
        // items [0, ncand): scores; [ncand, 5 ncand): coordinates
        const bool is_score = q < ncand;
        const int c = is_score ? q : (q - ncand) >> 2, k = (q - ncand) & 3;
        const int row = (int)(((unsigned)c * magicA) >> 16), an = c - row * g.A;
        const int4 ri = *reinterpret_cast<const int4*>(rowinfo + row * 4);
        const int i = ri.x + an;
        const float* rp = raw + row * LD;
        if (is_score) {
            const int cw = st_cls[c];
            const float sc = (cw < 0) ? st_sum[c] : 1.0f / st_sum[c] * sigmoid_f(rp[an]);
            scores[i] = sc;
            cls[i] = (sc >= conf_thresh) ? (cw & 0x7fffffff) : -1;
        } else {
            const float* tb = rp + g.A * (1 + g.C) + an * 4;
            boxes[(size_t)i * 4 + k] = decode_coord(g, scale, ri.y, ri.z, an, tb[k & 1], tb[2 + (k & 1)], (float)g.S, k);
        }
    }
}

template <int NT, int KMAX>
__global__ __launch_bounds__(256) void head_decode_kernel(GemmArgs a, GridInfo g, int scale, float conf_thresh,
                                                           float* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ cls, int dbg)
{
    __shared__ __attribute__((aligned(16))) c3h16 smem[HeadDecodeLds<NT>::HALVES];
    head_decode_block<NT, KMAX>(a, g, scale, conf_thresh, boxes, scores, cls, dbg, smem, blockIdx.x, gridDim.x);
}

// the three scales' last conv + decode as one launch (problem p = scale p)
template <int NT, int KMAX>
__global__ __launch_bounds__(256) void head_decode_group_kernel(Group<GemmArgs> q, GridInfo g, float conf_thresh,
                                                                 float* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ cls)
{
    __shared__ __attribute__((aligned(16))) c3h16 smem[HeadDecodeLds<NT>::HALVES];
    unsigned local, nb;
    const int p = group_problem(q.first, blockIdx.x, local, nb);
    head_decode_block<NT, KMAX>(q.a[p], g, p, conf_thresh, boxes, scores, cls, 0, smem, local, nb);
}

template <bool FULL>
static void launch_decode(const float* const heads[3], const GridInfo& g, int B, float conf_thresh,
                          float* boxes, float* scores, int32_t* cls, float* all_class, hipStream_t s)
{
    const long cands = (long)B * g.N;
    long blocks = (cands + 15) / 16;                        // 16 candidates per 256-thread block
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), blk(256);
    if (g.C <= 16) hipLaunchKernelGGL((decode_kernel<FULL, 1>), grid, blk, 0, s, heads[0], heads[1], heads[2], g, B, conf_thresh, boxes, scores, cls, all_class);
    else if (g.C <= 32) hipLaunchKernelGGL((decode_kernel<FULL, 2>), grid, blk, 0, s, heads[0], heads[1], heads[2], g, B, conf_thresh, boxes, scores, cls, all_class);
    else if (g.C <= 80) hipLaunchKernelGGL((decode_kernel<FULL, 5>), grid, blk, 0, s, heads[0], heads[1], heads[2], g, B, conf_thresh, boxes, scores, cls, all_class);
    else if (g.C <= 256) hipLaunchKernelGGL((decode_kernel<FULL, 16>), grid, blk, 0, s, heads[0], heads[1], heads[2], g, B, conf_thresh, boxes, scores, cls, all_class);
    else hipLaunchKernelGGL((decode_kernel<FULL, 64>), grid, blk, 0, s, heads[0], heads[1], heads[2], g, B, conf_thresh, boxes, scores, cls, all_class);
}

void launch_score_full(const float* const heads[3], const GridInfo& g, int B, float* all_bbox, float* all_class, hipStream_t s)
{
    launch_decode<true>(heads, g, B, 0.0f, all_bbox, nullptr, nullptr, all_class, s);
}

void launch_decode_cand(const float* const heads[3], const GridInfo& g, int B, float conf_thresh,
                        float* boxes, float* scores, int32_t* cls, hipStream_t s)
{
    launch_decode<false>(heads, g, B, conf_thresh, boxes, scores, cls, nullptr, s);
}

bool head_decode_supported(const GemmArgs& a, const GridInfo& g)
{
    return a.Wsh && a.Wsl && !a.pass && a.act == 0 && a.Npad <= 256 && g.A * (5 + g.C) <= a.Npad && g.C <= 80 && g.A <= 8 && a.M > 0;
}

void launch_head_decode(const GemmArgs& a, const GridInfo& g, int scale, float conf_thresh,
                        float* boxes, float* scores, int32_t* cls, hipStream_t s)
{
    const unsigned blocks = (unsigned)(((a.M + 31) / 32 + 7) & ~7);
    const dim3 grid(blocks), blk(256);
    static const int dbg = getenv("YN_HD_DBG") ? atoi(getenv("YN_HD_DBG")) : 0;
#define YN_HD(nt, kmax) hipLaunchKernelGGL((head_decode_kernel<nt, kmax>), grid, blk, 0, s, a, g, scale, conf_thresh, boxes, scores, cls, dbg)
    if (a.Npad > 128) YN_HD(2, 5);
    else if (g.C <= 16) YN_HD(1, 1);
    else if (g.C <= 32) YN_HD(1, 2);
    else YN_HD(1, 5);
#undef YN_HD
}

void launch_head_decode_group(const GemmArgs* a, int n, const GridInfo& g, float conf_thresh, float* boxes, float* scores, int32_t* cls, hipStream_t s)
{
    Group<GemmArgs> q{};
    unsigned tot = 0;
    for (int p = 0; p < YN_GROUP_MAX; ++p) {
        q.first[p] = tot;
        if (p < n) { q.a[p] = a[p]; tot += (unsigned)(((a[p].M + 31) / 32 + 7) & ~7); }
    }
    q.first[YN_GROUP_MAX] = tot;
    const dim3 grid(tot), blk(256);
#define YN_HDG(nt, kmax) hipLaunchKernelGGL((head_decode_group_kernel<nt, kmax>), grid, blk, 0, s, q, g, conf_thresh, boxes, scores, cls)
    if (a[0].Npad > 128) YN_HDG(2, 5);
    else if (g.C <= 16) YN_HDG(1, 1);
    else if (g.C <= 32) YN_HDG(1, 2);
    else YN_HDG(1, 5);
#undef YN_HDG
}

// -------------------------------------------------------------------------------------------------
// The tail of a detection head in ONE kernel (models/yolo_nano.py:60-82, 299-330, 362-367): depthwise 3x3 + pointwise conv (layers .2 and
// .3: dwpw_group_kernel's tile and thread roles), the last conv (.4) on the tile while it is still in LDS, and the candidate decode of
// head_decode_kernel.  Against dwpw_group_kernel + head_decode_group_kernel the 96-channel activation of layer .3 (44 MB per 32-image step
// at 416 x 416) is neither written nor read back, and a chip-filling launch disappears.  Workgroup = an 8 x 4 pixel tile of one image.
//   1. depthwise windows, taps, biases, the 96 x 96 split weight matrix and the first weight chunk of the last conv: one batch of loads
//   2. depthwise -> split planes A [32][104] x 2; W -> LDS; three wavefronts run the 96 -> 96 GEMM (K in gemm_split_tile's order)
//   3. barrier; activation, split -> the SAME planes (now the last conv's A operand); last conv's chunk 0 -> the weight space
//   4. four wavefronts x 64 columns: K = 96 in three chunks of 32 through one LDS buffer (register prefetch), as gemm_split_tile<1,4,2>
//   5. raw head tile [32][260] fp32 over the weight space, per-row candidate index / cell, then head_decode_kernel's three decode passes
// Every sum runs in the order of the separate kernels and the operand split is the same function of the same fp32 values: bit-identical
// outputs (test_head_tail_is_bit_identical).  LDS 50 KB: three workgroups per CU.
// -------------------------------------------------------------------------------------------------
template <int KMAX>
__device__ __forceinline__ void head_tail_block(const HeadTailArgs& a, const GridInfo& g, int scale, float conf_thresh,
                                                float* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ cls,
                                                c3h16* smem, unsigned bid, unsigned nblocks)
{
    constexpr int TW = 8, TH = 4, NO = TW * TH, C = 96, KQ = C / 8, BN1 = 96, AST = C + 8, R = 4, BN2 = 256, LD = BN2 + 4;
    c3h16* Ah = smem;                                       // [NO][AST]
    c3h16* Al = Ah + NO * AST;
    c3h16* Bs = Al + NO * AST;                              // [2][KQ][BN1][8], then chunks [2][4][BN2][8], then the raw tile
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, h = lane >> 5;
    const int tx_n = (a.W + TW - 1) / TW, ty_n = (a.H + TH - 1) / TH;
    const int tile = (int)(((bid & 7u) * (nblocks >> 3)) + (bid >> 3));
    if (tile >= a.B * ty_n * tx_n) return;
    const int b = tile / (ty_n * tx_n), trem = tile - b * (ty_n * tx_n);
    const int oy0 = (trem / tx_n) * TH, ox0 = (trem % tx_n) * TW;

#ifdef YN_EXP_TIMING
    long long TS[10]; int tsn = 0;
#define YN_TS() TS[tsn++] = __builtin_readcyclecounter()
#else
#define YN_TS()
#endif
    YN_TS();
    // ---- 1. loads -----------------------------------------------------------------------------------------------------------------
    // the 96 x 96 split weights go straight into LDS: 2 304 sixteen-byte LDS-DMA pieces, nine per thread, no registers (round 5; they were nine
    // 16-byte register loads per thread, staged by hand after the depthwise phase: 36 registers live across it, and with the windows and the
    // taps the kernel's budget was exceeded - hipcc issued the batch in three instalments, each a memory round trip)
    {
        constexpr int NP1 = 2 * KQ * BN1;                   // pieces: plane (hi / lo), octet, column
        static_assert(NP1 % 256 == 0 && (KQ * BN1) % 64 == 0, "whole wavefronts per plane");
        const unsigned lds_bs = (unsigned)(size_t)(__attribute__((address_space(3))) c3h16*)Bs;
#pragma unroll
        for (int i = 0; i < NP1 / 256; ++i) {
            const int g0 = 256 * i + 64 * wave;             // this wavefront's first piece (wave-uniform)
            const int pl = g0 / (KQ * BN1);
            dma16(pl ? a.Wl : a.Wh, (unsigned)(g0 - pl * (KQ * BN1) + lane) * 16u, lds_bs + (unsigned)g0 * 16u);
        }
    }
    const int cq = t % (C / 4), run = t / (C / 4);          // 24 channel quads x 8 runs of 4 pixels = 192 workers
    const bool worker = run < 8;
    const int c = cq * 4, ry = run >> 1, rx = (run & 1) * R;
    float4 win[3][R + 2], wd[9], bd = make_float4(0.f, 0.f, 0.f, 0.f);
    if (worker) {
        const int oy = oy0 + ry;
        unsigned wok = 0;                                   // which window values are inside the image: the masks are applied AFTER the whole batch
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy - 1 + ky;
            const bool yok = oy < a.H && iy >= 0 && iy < a.H;
            const float* rowp = a.in + ((size_t)(b * a.H + (yok ? iy : 0)) * a.W) * C + c;
#pragma unroll
            for (int j = 0; j < R + 2; ++j) {
                const int ix = ox0 + rx - 1 + j;
                const bool ok = yok && ix >= 0 && ix < a.W;
                win[ky][j] = *reinterpret_cast<const float4*>(rowp + (size_t)(ok ? ix : 0) * C);
                wok |= (ok ? 1u : 0u) << (ky * (R + 2) + j);
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) wd[k] = *reinterpret_cast<const float4*>(a.wdw + k * C + c);
        bd = *reinterpret_cast<const float4*>(a.bdw + c);
        __builtin_amdgcn_sched_barrier(0);                  // all 28 loads are issued before the first is used (one memory round trip, not three)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int j = 0; j < R + 2; ++j) win[ky][j] = vmask(win[ky][j], 0u - ((wok >> (ky * (R + 2) + j)) & 1u));
    }
    const float gbias = (wave < 3) ? a.bias[wave * 32 + l31] : 0.0f;
    constexpr int B2_PER = 2 * 4 * BN2 / 256;               // 8 granules per thread and chunk
    c3h16x8 b2_reg[B2_PER];
    auto prefetch_b2 = [&](int chunk) {
#pragma unroll
        for (int i = 0; i < B2_PER; ++i) {
            const int gi = t + 256 * i;
            const int pl = gi / (4 * BN2), r = gi - pl * (4 * BN2);
            const int o = r / BN2, n = r - o * BN2;
            // (no select on the loaded value: it would be a USE where the load is issued, and hipcc then waits for every load of the prefetch
            // in turn, inside the MFMA section it is meant to hide under.  A column past Npad reads column Npad - 1: never decoded.)
            b2_reg[i] = *reinterpret_cast<const c3h16x8*>(reinterpret_cast<const c3h16*>(pl ? a.Wfl : a.Wfh) + ((size_t)(chunk * 4 + o) * a.Npad + (n < a.Npad ? n : a.Npad - 1)) * 8);
        }
    };
    auto stage_b2 = [&]() {
#pragma unroll
        for (int i = 0; i < B2_PER; ++i) *reinterpret_cast<c3h16x8*>(Bs + (size_t)(t + 256 * i) * 8) = b2_reg[i];
    };
    float fbias[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) { const int n = wave * 64 + nt * 32 + l31; fbias[nt] = a.fbias[n < a.Npad ? n : 0]; }

    // ---- 2. depthwise -> split planes; W -> LDS; 96 -> 96 GEMM on three wavefronts -----------------------------------------------------
    float amax = 0.0f;                                      // range guard (yn_device.h): largest |value| this thread has split
    if (worker) {
#pragma unroll
        for (int o = 0; o < R; ++o) {
            float4 acc = bd;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) vfma(acc, win[ky][o + kx], wd[ky * 3 + kx]);
            acc = vact(acc, a.dw_act);
            const int op = ry * TW + rx + o;
            const float x4[4] = {acc.x, acc.y, acc.z, acc.w};
            c3h16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) { amax = range_track(amax, x4[j]); hi[j] = (c3h16)x4[j]; lo[j] = (c3h16)((x4[j] - (float)hi[j]) * 2048.0f); }
            *reinterpret_cast<c3h16x4*>(Ah + op * AST + c) = hi;
            *reinterpret_cast<c3h16x4*>(Al + op * AST + c) = lo;
        }
    }
    vm_drain();                                             // the weight pieces (issued a depthwise phase ago)
    prefetch_b2(0);
    __syncthreads();
    YN_TS();
    f32x16 m0, m1;
#pragma unroll
    for (int k = 0; k < 16; ++k) { m0[k] = 0.0f; m1[k] = 0.0f; }
    if (wave < 3) {
        const c3h16* Ahb = Ah + l31 * AST + h * 8;
        const c3h16* Alb = Al + l31 * AST + h * 8;
        const c3h16* Bhb = Bs + (size_t)(h * BN1 + wave * 32 + l31) * 8;
        const c3h16* Blb = Bhb + (size_t)KQ * BN1 * 8;
#pragma unroll
        for (int ks = 0; ks < KQ / 2; ++ks) {
            const c3h16x8 ah = *reinterpret_cast<const c3h16x8*>(Ahb + ks * 16);
            const c3h16x8 al = *reinterpret_cast<const c3h16x8*>(Alb + ks * 16);
            const c3h16x8 bh = *reinterpret_cast<const c3h16x8*>(Bhb + (size_t)(ks * 2 * BN1) * 8);
            const c3h16x8 bl = *reinterpret_cast<const c3h16x8*>(Blb + (size_t)(ks * 2 * BN1) * 8);
            m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, m1, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, m1, 0, 0, 0);
        }
    }
    __syncthreads();                                        // every operand read of the first GEMM is done: planes and weight space are free
    YN_TS();

    // ---- 3. layer .3's output tile -> the planes (the last conv's A operand); its first weight chunk -> LDS ------------------------------
    if (wave < 3) {
        const int n = wave * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const float v = apply_act(__builtin_fmaf(m1[r], 1.0f / 2048.0f, m0[r]) + gbias, a.act);
            amax = range_track(amax, v);
            const c3h16 hi = (c3h16)v;
            Ah[row * AST + n] = hi;
            Al[row * AST + n] = (c3h16)((v - (float)hi) * 2048.0f);
        }
    }
    range_report(a.ovf, amax);                              // both split sites of this workgroup are behind it
    stage_b2();
    __syncthreads();
    YN_TS();
    prefetch_b2(1);

    // ---- 4. the last conv: 32 x 256, wavefront = 64 columns ---------------------------------------------------------------------------
    f32x16 acc0[2], acc1[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc0[nt][k] = 0.0f; acc1[nt][k] = 0.0f; }
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        const c3h16* Ahb = Ah + l31 * AST + ch * 32 + h * 8;
        const c3h16* Alb = Al + l31 * AST + ch * 32 + h * 8;
        const c3h16* Bhb = Bs + (size_t)(h * BN2 + wave * 64 + l31) * 8;
        const c3h16* Blb = Bhb + (size_t)4 * BN2 * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const c3h16x8 ah = *reinterpret_cast<const c3h16x8*>(Ahb + ks * 16);
            const c3h16x8 al = *reinterpret_cast<const c3h16x8*>(Alb + ks * 16);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                const c3h16x8 bh = *reinterpret_cast<const c3h16x8*>(Bhb + (size_t)(ks * 2 * BN2 + nt * 32) * 8);
                const c3h16x8 bl = *reinterpret_cast<const c3h16x8*>(Blb + (size_t)(ks * 2 * BN2 + nt * 32) * 8);
                acc0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0[nt], 0, 0, 0);
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[nt], 0, 0, 0);
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[nt], 0, 0, 0);
            }
        }
        __syncthreads();                                    // every wavefront is done with this chunk's weights (and, after the last one, with the planes)
        if (ch + 1 < 3) {
            stage_b2();
            __syncthreads();
            if (ch + 2 < 3) prefetch_b2(ch + 2);
        }
    }

    YN_TS();
    // ---- 5. raw tile, per-row candidate index, decode ------------------------------------------------------------------------------------
    float* raw = reinterpret_cast<float*>(Bs);
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int n = wave * 64 + nt * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            raw[((r & 3) + 8 * (r >> 2) + 4 * h) * LD + n] = __builtin_fmaf(acc1[nt][r], 1.0f / 2048.0f, acc0[nt][r]) + fbias[nt];
    }
    int* rowinfo = reinterpret_cast<int*>(raw + NO * LD + NO * 16);     // [32][4]: first candidate, gx, gy, inside the image
    if (t < NO) {
        const int py = oy0 + t / TW, px = ox0 + t % TW;
        const bool ok = py < a.H && px < a.W;
        const int cell = ok ? py * a.W + px : 0;
        *reinterpret_cast<int4*>(rowinfo + t * 4) = make_int4(b * g.N + g.off[scale] + cell * g.A, px, py, ok ? 1 : 0);
    }
    const unsigned magicA = (65536u + (unsigned)g.A - 1u) / (unsigned)g.A;     // c / A == (c * magicA) >> 16 for c < 32 * A <= 256
    __syncthreads();
    YN_TS();
    float* st_sum = raw + NO * LD;                          // [32 * A] sum of exponentials, or the final score (general path)
    int* st_cls = reinterpret_cast<int*>(st_sum + NO * g.A);
    const int j = t & 15;
    const int ncand = NO * g.A;
    const float skip_logit = conf_skip_logit(conf_thresh);
    // Three candidates per 16-lane group and round (round 5): a candidate is ONE dependency chain - LDS read, max, four row steps, five
    // exponentials, sum, four row steps, the wave-level decision, four row steps - and at three wavefronts per SIMD a group's six candidates one
    // after the other were 7.5 k of a workgroup's 27 k cycles.  The branch-free halves of three candidates run back to back; the decisions
    // follow per candidate, on the same four candidates per wavefront as before (candidate = group + 16 x (3 x round + u)): the same bits.
    constexpr int CU = 3;
    for (int c0 = t >> 4; c0 < ncand; c0 += 16 * CU) {     // (ncand = 32 A: whole wavefronts)
        const float* rp[CU];
        int an[CU];
        float obj[CU];
        bool inr[CU], below[CU];
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int cnd = c0 + 16 * u;
            inr[u] = cnd < ncand;
            const int cc = inr[u] ? cnd : c0;
            const int row = (int)(((unsigned)cc * magicA) >> 16);
            an[u] = cc - row * g.A;
            rp[u] = raw + row * LD;
            obj[u] = rp[u][an[u]];
        }
        bool all_below = true;
#pragma unroll
        for (int u = 0; u < CU; ++u) { below[u] = inr[u] && wave_below_conf(obj[u], skip_logit); all_below = all_below && (below[u] || !inr[u]); }
        float v[CU][KMAX], sum[CU];
        int first[CU];
        bool general[CU];
        if (!all_below) {
#pragma unroll
            for (int u = 0; u < CU; ++u) cand_class_stats<false, KMAX>(g, rp[u], an[u], j, obj[u], v[u], sum[u], first[u], general[u]);
        }
#pragma unroll
        for (int u = 0; u < CU; ++u) {
            const int cnd = c0 + 16 * u;
            if (!inr[u]) continue;
            if (below[u]) { if (j == 0) { st_sum[cnd] = 0.0f; st_cls[cnd] = (int)0x80000000; } continue; }
            float sc;
            int cbest;
            const bool fin = cand_class_finish<false, KMAX>(g, 0, j, obj[u], v[u], sum[u], first[u], general[u], sc, cbest, nullptr);
            if (j == 0) { st_sum[cnd] = fin ? sc : sum[u]; st_cls[cnd] = fin ? (cbest | (int)0x80000000) : cbest; }
        }
    }
    __syncthreads();
    YN_TS();
    for (int q = t; q < 5 * ncand; q += 256) {
        const bool is_score = q < ncand;
        const int cnd = is_score ? q : (q - ncand) >> 2, k = (q - ncand) & 3;
        const int row = (int)(((unsigned)cnd * magicA) >> 16), an = cnd - row * g.A;
        const int4 ri = *reinterpret_cast<const int4*>(rowinfo + row * 4);
        if (!ri.w) continue;                                // tile pixel outside the image
        const int i = ri.x + an;
        const float* rp = raw + row * LD;
        if (is_score) {
            const int cw = st_cls[cnd];
            const float sc = (cw < 0) ? st_sum[cnd] : 1.0f / st_sum[cnd] * sigmoid_f(rp[an]);
            scores[i] = sc;
            cls[i] = (sc >= conf_thresh) ? (cw & 0x7fffffff) : -1;
        } else {
            const float* tb = rp + g.A * (1 + g.C) + an * 4;
            boxes[(size_t)i * 4 + k] = decode_coord(g, scale, ri.y, ri.z, an, tb[k & 1], tb[2 + (k & 1)], (float)g.S, k);
        }
    }
#ifdef YN_EXP_TIMING
    YN_TS();
    if (t == 0 && scale == 0 && (bid % 401) == 7)
        printf("headtail blk %u loads+dw %lld gemm1 %lld split+stage %lld gemm2 %lld rawtile %lld passA %lld passBC %lld total %lld\n", bid, TS[1] - TS[0], TS[2] - TS[1], TS[3] - TS[2],
               TS[4] - TS[3], TS[5] - TS[4], TS[6] - TS[5], TS[7] - TS[6], TS[7] - TS[0]);
#endif
#undef YN_TS
}

constexpr int HEAD_TAIL_HALVES = 2 * 32 * 104 + 2 * 12 * 96 * 8;        // planes + the larger of {W 96x96, a chunk of the last conv, the raw tile}
static_assert(2 * 12 * 96 * 8 >= 2 * 4 * 256 * 8 && 2 * 12 * 96 * 8 >= 32 * 260 * 2 + 32 * 8 * 4 + 32 * 4 * 2, "weight space holds a chunk and the raw tile");

template <int KMAX>
__global__ __attribute__((amdgpu_flat_work_group_size(256, 256), amdgpu_waves_per_eu(3, 3)))     // exactly three wavefronts per SIMD: LDS holds three workgroups; aiming at four costs the load batch its registers
void head_tail_group_kernel(Group<HeadTailArgs> q, GridInfo g, float conf_thresh,
                                                                  float* __restrict__ boxes, float* __restrict__ scores, int32_t* __restrict__ cls)
{
    extern __shared__ __attribute__((aligned(16))) float head_tail_smem[];
    unsigned local, nb;
    const int p = group_problem(q.first, blockIdx.x, local, nb);
    head_tail_block<KMAX>(q.a[p], g, p, conf_thresh, boxes, scores, cls, reinterpret_cast<c3h16*>(head_tail_smem), local, nb);
}

bool head_tail_ok(const HeadTailArgs* q, int n, const GridInfo& g)
{
    if (n < 1 || n > YN_GROUP_MAX || g.C > 80 || g.C <= 32 || g.A > 8 || g.A < 1) return false;
    for (int p = 0; p < n; ++p)
        if (!q[p].Wh || !q[p].Wl || !q[p].Wfh || !q[p].Wfl || q[p].Npad <= 128 || q[p].Npad > 256 || g.A * (5 + g.C) > q[p].Npad || q[p].B <= 0) return false;
    return true;
}

void launch_head_tail_group(const HeadTailArgs* a, int n, const GridInfo& g, float conf_thresh, float* boxes, float* scores, int32_t* cls, hipStream_t s)
{
    Group<HeadTailArgs> q{};
    unsigned tot = 0;
    for (int p = 0; p < YN_GROUP_MAX; ++p) {
        q.first[p] = tot;
        if (p < n) { q.a[p] = a[p]; tot += (unsigned)(((unsigned)a[p].B * ((a[p].H + 3) / 4) * ((a[p].W + 7) / 8) + 7u) & ~7u); }
    }
    q.first[YN_GROUP_MAX] = tot;
    const size_t lds = (size_t)HEAD_TAIL_HALVES * 2;
    static unsigned long long attr = 0;
    if (attr_pending(attr)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(head_tail_group_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipLaunchKernelGGL((head_tail_group_kernel<5>), dim3(tot), dim3(256), lds, s, q, g, conf_thresh, boxes, scores, cls);
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

// =================================================================================================
// Per-class greedy NMS (models/yolo_nano.py:159-188, 263-272), exact, in five kernels:
//   bucket   per image: class histogram -> segment offsets, scatter candidate ids, tile offsets
//   sort     per (image, class) segment: bitonic sort of (score, id) keys, descending
//            == the reference's `scores.argsort()[::-1]` with the tie rule "higher index first";
//            gathers the boxes into sorted order
//   matrix   all segments, all 64x64 tiles (row chunk ri <= col chunk ci) in parallel over the whole
//            chip: bit (r, t) = "sorted item r suppresses sorted item t" (float32 arithmetic == numpy's)
//   resolve  one wavefront per segment walks the chunks in order: 64-step scalar resolve of the
//            diagonal tile, then ORs the kept rows' words into the removed mask of later chunks
//   compact  kept candidates in ascending candidate order (:274-277)
// The O(n^2) IoU work is spread over all CUs; the serial part is n/64 short rounds per segment.
// =================================================================================================
typedef unsigned long long u64;

__host__ __device__ inline int nms_pow2(int n) { int p = 64; while (p < n) p <<= 1; return p; }

__global__ __launch_bounds__(1024) void bucket_kernel(const int32_t* __restrict__ cls, int N, int C,
                                                       int32_t* __restrict__ seg_count, int32_t* __restrict__ seg_off,
                                                       int32_t* __restrict__ tile_off, int32_t* __restrict__ bucket,
                                                       int32_t* __restrict__ keep, int32_t* __restrict__ large_list,
                                                       int large_cap, int large_thresh, int32_t* __restrict__ seg_order)
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
        int run = 0, tiles = 0, nlarge = 0;
        for (int c = 0; c < C; ++c) {
            const int h = hist[c];
            seg_count[(size_t)b * C + c] = h;
            seg_off[(size_t)b * C + c] = run;
            tile_off[(size_t)b * (C + 1) + c] = tiles;
            cursor[c] = run;
            run += h;
            const int T = (h + 63) >> 6;
            tiles += T * (T + 1) / 2;
            if (h > large_thresh && nlarge < large_cap) large_list[(size_t)b * (large_cap + 1) + 1 + nlarge++] = c;
        }
        tile_off[(size_t)b * (C + 1) + C] = tiles;
        large_list[(size_t)b * (large_cap + 1)] = nlarge;
    }
    // the image's classes by segment size, largest first (seg_order[rank] = class): per-segment kernels whose workgroups take as long
    // as their segment is large start the long ones FIRST instead of wherever the class id puts them (C is small: rank by counting)
    if (seg_order)
        for (int c = threadIdx.x; c < C; c += blockDim.x) {
            const int h = hist[c];
            int rank = 0;
            for (int o = 0; o < C; ++o) { const int ho = hist[o]; rank += (ho > h || (ho == h && o < c)) ? 1 : 0; }
            seg_order[(size_t)b * C + rank] = c;
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

__device__ __forceinline__ unsigned order_bits(float f)
{
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);     // unsigned order == float order
}

// true when box j must be REMOVED given kept box i  (reference keeps `ovr <= thresh`; NaN -> removed)
__device__ __forceinline__ bool suppressed(const float4 bi, float ai, const float4 bj, float aj, float thresh, int diou)
{
    const float xx1 = fmaxf(bi.x, bj.x), yy1 = fmaxf(bi.y, bj.y);
    const float xx2 = fminf(bi.z, bj.z), yy2 = fminf(bi.w, bj.w);
    const float dw = xx2 - xx1, dh = yy2 - yy1;
    const float t0 = ai + aj;
    const float w = fmaxf(1e-28f, dw);
    const float h = fmaxf(1e-28f, dh);
    const float inter = w * h;
    const float un = t0 - inter;
    if (!diou) {
        // Division-free early outs that cannot change the result: rounding is monotonic, so
        // inter/un < thresh (real arithmetic, with a 1e-5 guard band for the rounding of thresh*un)
        // implies fl(inter/un) <= thresh, and inter/un > thresh*(1+1e-5) implies fl(inter/un) > thresh.
        const float p = thresh * un;
        if (un > 0.0f && p > 1e-30f) {
            if (inter < p * 0.99999f) return false;
            if (inter > p * 1.00001f) return true;
        }
    }
    float ovr = inter / un;
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

// ---- sort ------------------------------------------------------------------------------------------
// keys live in LDS (KEYS_IN_LDS) or, for segments beyond the LDS capacity, in a global scratch that only
// this workgroup touches (agent-scope relaxed accesses so that the CU's L1 never serves a stale key).
template <bool KEYS_IN_LDS>
__device__ __forceinline__ u64 key_load(u64* k, int i)
{
    if (KEYS_IN_LDS) return k[i];
    return __hip_atomic_load(k + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <bool KEYS_IN_LDS>
__device__ __forceinline__ void key_store(u64* k, int i, u64 v)
{
    if (KEYS_IN_LDS) k[i] = v;
    else __hip_atomic_store(k + i, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

template <bool KEYS_IN_LDS>
__device__ void sort_segment(const float* __restrict__ boxes, const float* __restrict__ scores, int32_t* __restrict__ ids, bool ids_valid,
                             int n, int P, u64* keys, float4* __restrict__ sbox, u64* __restrict__ key_out = nullptr)
{
    const int tid = threadIdx.x, nthr = blockDim.x;
    for (int j = tid; j < P; j += nthr) {
        u64 k = 0;                                          // padding sorts to the end
        if (j < n) {
            const int id = ids_valid ? ids[j] : j;
            k = ((u64)order_bits(scores[id]) << 32) | (unsigned)id;
        }
        key_store<KEYS_IN_LDS>(keys, j, k);
    }
    __syncthreads();
    if (KEYS_IN_LDS) {
        // The three innermost sub-stages of every phase (partners 4, 2, 1 apart) and the whole of phases 2, 4, 8 work on 8 consecutive keys:
        // a thread takes them into registers (four 16-byte LDS reads), runs the compare-exchanges there and writes them back — one LDS round
        // trip instead of three (six for the first phases); 33 of the 78 sub-stages of a 4 096-key sort.  Same comparator, same network
        // order inside a phase: the sorted result is the unique descending order of the (score, id) keys either way.
        auto cx = [](u64& x, u64& y, bool desc) { if ((x < y) == desc) { const u64 t2 = x; x = y; y = t2; } };
        auto load8 = [&](int g, u64 (&v)[8]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const ulonglong2 w = *reinterpret_cast<const ulonglong2*>(keys + 8 * g + 2 * q);
                v[2 * q] = w.x; v[2 * q + 1] = w.y;
            }
        };
        auto store8 = [&](int g, const u64 (&v)[8]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) *reinterpret_cast<ulonglong2*>(keys + 8 * g + 2 * q) = make_ulonglong2(v[2 * q], v[2 * q + 1]);
        };
        auto tail3 = [&](u64 (&v)[8], bool desc) {           // partners 4, 2, 1 apart, one direction for the group
            cx(v[0], v[4], desc); cx(v[1], v[5], desc); cx(v[2], v[6], desc); cx(v[3], v[7], desc);
            cx(v[0], v[2], desc); cx(v[1], v[3], desc); cx(v[4], v[6], desc); cx(v[5], v[7], desc);
            cx(v[0], v[1], desc); cx(v[2], v[3], desc); cx(v[4], v[5], desc); cx(v[6], v[7], desc);
        };
        for (int g = tid; g < (P >> 3); g += nthr) {
            u64 v[8];
            load8(g, v);
            cx(v[0], v[1], true); cx(v[2], v[3], false); cx(v[4], v[5], true); cx(v[6], v[7], false);       // k = 2: descending where (index & 2) == 0
            cx(v[0], v[2], true); cx(v[1], v[3], true); cx(v[4], v[6], false); cx(v[5], v[7], false);       // k = 4: (index & 4) == 0
            cx(v[0], v[1], true); cx(v[2], v[3], true); cx(v[4], v[5], false); cx(v[6], v[7], false);
            tail3(v, ((8 * g) & 8) == 0);                                                                      // k = 8
            store8(g, v);
        }
        __syncthreads();
        for (int k = 16; k <= P; k <<= 1) {
            for (int j = k >> 1; j >= 8; j >>= 1) {
                for (int i = tid; i < (P >> 1); i += nthr) {
                    const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                    const int b = a + j;
                    const u64 x = keys[a], y = keys[b];
                    const bool desc = (a & k) == 0;
                    if ((x < y) == desc) { keys[a] = y; keys[b] = x; }
                }
                // pairs with j < 64 stay inside the 128-key block this wavefront also owned in the previous sub-stage (i -> block i/64),
                // and a wavefront's LDS accesses are executed in order: no workgroup barrier until the register pass changes the ownership
                if (j > 8 && j <= 64) __builtin_amdgcn_wave_barrier();
                else __syncthreads();
            }
            for (int g = tid; g < (P >> 3); g += nthr) {
                u64 v[8];
                load8(g, v);
                tail3(v, ((8 * g) & k) == 0);
                store8(g, v);
            }
            __syncthreads();
        }
    } else {
    for (int k = 2; k <= P; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int i = tid; i < (P >> 1); i += nthr) {
                const int a = ((i & ~(j - 1)) << 1) | (i & (j - 1));
                const int b = a + j;
                const u64 x = key_load<KEYS_IN_LDS>(keys, a), y = key_load<KEYS_IN_LDS>(keys, b);
                const bool desc = (a & k) == 0;
                if ((x < y) == desc) { key_store<KEYS_IN_LDS>(keys, a, y); key_store<KEYS_IN_LDS>(keys, b, x); }
            }
            __syncthreads();
        }
    }
    }
    if (key_out) {                                          // a CHUNK of a large segment: its sorted keys, for sort_merge_kernel
        for (int j = tid; j < n; j += nthr) key_out[j] = key_load<KEYS_IN_LDS>(keys, j);
        return;
    }
    for (int j = tid; j < n; j += nthr) {
        const int id = (int)(unsigned)(key_load<KEYS_IN_LDS>(keys, j) & 0xffffffffu);
        ids[j] = id;
        sbox[j] = *reinterpret_cast<const float4*>(boxes + (size_t)id * 4);
    }
}

#define YN_SORT_SMALL 1024
#define YN_SORT_LARGE 16384

// grid (C, B): segments with n_lo < n <= n_hi
__global__ __launch_bounds__(1024) void sort_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                    const int32_t* __restrict__ seg_count, const int32_t* __restrict__ seg_off,
                                                    int32_t* __restrict__ bucket, float4* __restrict__ sbox, int N, int C,
                                                    int n_lo, int n_hi, u64* __restrict__ gscratch, size_t gscratch_stride,
                                                    const int32_t* __restrict__ large_list, int large_cap, const int32_t* __restrict__ seg_order)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    int b = blockIdx.y;
    int c = blockIdx.x;
    if (large_list) {                                       // grid.x indexes the image's list of large segments
        const int32_t* ll = large_list + (size_t)b * (large_cap + 1);
        if (c >= ll[0]) return;
        c = ll[1 + c];
    } else if (seg_order) {                                 // grid (B, C): largest segments of every image first (bucket_kernel's order)
        b = blockIdx.x;
        c = seg_order[(size_t)b * C + blockIdx.y];
    }
    const int n = seg_count[(size_t)b * C + c];
    if (n <= n_lo || n > n_hi) return;
    const int off = seg_off[(size_t)b * C + c];
    const int P = nms_pow2(n);
    if (gscratch)
        sort_segment<false>(boxes + (size_t)b * N * 4, scores + (size_t)b * N, bucket + (size_t)b * N + off, true, n, P,
                            gscratch + (size_t)b * gscratch_stride, sbox + (size_t)b * N + off);
    else
        sort_segment<true>(boxes + (size_t)b * N * 4, scores + (size_t)b * N, bucket + (size_t)b * N + off, true, n, P,
                           reinterpret_cast<u64*>(sort_lds), sbox + (size_t)b * N + off);
}

// ---- large segments without large workgroups (round 5) -------------------------------------------------------------------------------
// A segment above YN_SORT_SMALL boxes used to get a 1024-thread workgroup with 128 KB of LDS (bitonic network over the whole segment, 26 us
// for the benchmark's 4 300-box class) - a workgroup that needs a nearly EMPTY CU to start: alone that is a 32-workgroup launch on an idle
// chip; with other batches in flight (bench.py's four streams) every one of the launch's 352 workgroups, most of which only look at the list and
// leave, waits for a CU to drain.  Dropping the sort from the four-stream run saved 53 us of a 710 us step - as much as the kernel takes alone.
// Now: the segment's 1024-box CHUNKS are sorted by the same 256-thread / 8 KB workgroups as the small segments, IN the same launch (extra grid
// rows: the image's chunk slots, enumerated from bucket_kernel's list), keys out to scratch (the not yet used matrix area); sort_merge_kernel
// then places every box: its final position = the number of keys above it in every chunk of its segment, eleven-step branch-free binary
// searches, all chunks of a group in flight together.  Keys are unique ((score, id)), so the order is the one the network produced.
#define YN_SORT_CHUNK 1024
#define YN_SORT_CHUNK_LOG 10
__host__ __device__ inline int nms_chunk_slots(int N, int large_cap) { return N / YN_SORT_CHUNK + large_cap; }     // sum over large segments of ceil(n / 1024) <= N/1024 + their number

// grid (B, C + chunk slots), block 256, LDS 8 KB
__global__ __launch_bounds__(256) void sort_chunk_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                         const int32_t* __restrict__ seg_count, const int32_t* __restrict__ seg_off,
                                                         int32_t* __restrict__ bucket, float4* __restrict__ sbox, int N, int C,
                                                         const int32_t* __restrict__ seg_order, const int32_t* __restrict__ large_list, int large_cap,
                                                         u64* __restrict__ keys_g, size_t keys_stride)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    const int b = blockIdx.x;
    if ((int)blockIdx.y < C) {                              // a small segment, whole (largest of the image first)
        const int c = seg_order ? seg_order[(size_t)b * C + blockIdx.y] : (int)blockIdx.y;
        const int n = seg_count[(size_t)b * C + c];
        if (n <= 0 || n > YN_SORT_SMALL) return;
        const int off = seg_off[(size_t)b * C + c];
        sort_segment<true>(boxes + (size_t)b * N * 4, scores + (size_t)b * N, bucket + (size_t)b * N + off, true, n, nms_pow2(n),
                           reinterpret_cast<u64*>(sort_lds), sbox + (size_t)b * N + off);
        return;
    }
    // chunk slot -> (large segment, chunk): walk the image's list (at most N/1024 entries)
    int slot = (int)blockIdx.y - C;
    const int32_t* ll = large_list + (size_t)b * (large_cap + 1);
    const int nl = ll[0];
    int c = -1, n = 0;
    for (int l = 0; l < nl; ++l) {
        const int cl = ll[1 + l];
        const int nc = seg_count[(size_t)b * C + cl];
        const int chunks = (nc + YN_SORT_CHUNK - 1) >> YN_SORT_CHUNK_LOG;
        if (slot < chunks) { c = cl; n = nc; break; }
        slot -= chunks;
    }
    if (c < 0) return;
    const int j0 = slot << YN_SORT_CHUNK_LOG;
    const int len = min(YN_SORT_CHUNK, n - j0);
    const int off = seg_off[(size_t)b * C + c] + j0;
    sort_segment<true>(boxes + (size_t)b * N * 4, scores + (size_t)b * N, bucket + (size_t)b * N + off, true, len, nms_pow2(len),
                       reinterpret_cast<u64*>(sort_lds), nullptr, keys_g + (size_t)b * keys_stride + off);
}

// grid (ceil(N / 256), B), block 256: thread = one position of the image's class-grouped candidate array
__global__ __launch_bounds__(256) void sort_merge_kernel(const float* __restrict__ boxes, const int32_t* __restrict__ seg_count,
                                                         const int32_t* __restrict__ seg_off, int32_t* __restrict__ bucket, float4* __restrict__ sbox,
                                                         int N, int C, const int32_t* __restrict__ large_list, int large_cap,
                                                         const u64* __restrict__ keys_g, size_t keys_stride)
{
    const int b = blockIdx.y;
    const int pos = blockIdx.x * 256 + threadIdx.x;
    const int32_t* ll = large_list + (size_t)b * (large_cap + 1);
    const int nl = ll[0];
    int off = 0, n = 0;                                     // the large segment this position lies in (segments may lie in any order: scan the list)
    for (int l = 0; l < nl; ++l) {
        const int cl = ll[1 + l];
        const int o = seg_off[(size_t)b * C + cl], nc = seg_count[(size_t)b * C + cl];
        if (pos >= o && pos < o + nc) { off = o; n = nc; }
    }
    if (n == 0) return;
    const u64* K = keys_g + (size_t)b * keys_stride + off;
    const u64 key = K[pos - off];
    const int nch = (n + YN_SORT_CHUNK - 1) >> YN_SORT_CHUNK_LOG;
    int rank = 0;
    for (int o0 = 0; o0 < nch; o0 += 8) {                   // eight chunks' searches in flight together (sixteen: slower - the absent chunks' loads are issued too)
        int lo[8], len[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) { lo[u] = 0; len[u] = o0 + u < nch ? min(YN_SORT_CHUNK, n - ((o0 + u) << YN_SORT_CHUNK_LOG)) : 0; }
#pragma unroll
        for (int st = YN_SORT_CHUNK; st >= 1; st >>= 1) {   // largest lo with chunk[lo - 1] > key (descending chunk; the box's own chunk counts its position)
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int p2 = lo[u] + st;
                const int idx = ((o0 + u) << YN_SORT_CHUNK_LOG) + (p2 <= len[u] ? p2 - 1 : 0);
                const u64 kv = K[o0 + u < nch ? idx : 0];
                if (p2 <= len[u] && kv > key) lo[u] = p2;
            }
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) rank += lo[u];
    }
    const int id = (int)(unsigned)(key & 0xffffffffu);
    bucket[(size_t)b * N + off + rank] = id;
    sbox[(size_t)b * N + off + rank] = *reinterpret_cast<const float4*>(boxes + ((size_t)b * N + id) * 4);
}

// ---- bucket + sort in ONE launch, for the few-segment case (B * C <= 256: one to three images) ---------------------------------------
// bucket_kernel's histogram and scatter are LDS atomics, and with a skewed class distribution (the benchmark's random weights put a
// quarter of an image's 10 647 candidates into one class) they serialise on ONE address at ~16 cycles each: 25 us for a kernel that
// moves 40 KB - a fifth of the whole per-image NMS.  Here every (class, image) workgroup scans the image's class labels itself and
// takes its own members with ballots (no atomics, ascending candidate order), reserves its place in the id / sorted-box arrays with ONE
// atomic (segments may lie in any order: every consumer goes through seg_off), sorts (sort_segment), and the LAST workgroup of an image to
// finish writes the tile offsets - the one prefix over the classes the matrix / resolve kernels need - and resets the counters.
// ctr: [B][2] ints (position cursor, finished workgroups), zero on entry, zero on exit.
__global__ __launch_bounds__(1024) void bucket_sort_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, const int32_t* __restrict__ cls,
                                                           int N, int C, int32_t* __restrict__ seg_count, int32_t* __restrict__ seg_off,
                                                           int32_t* __restrict__ tile_off, int32_t* __restrict__ bucket, int32_t* __restrict__ keep,
                                                           float4* __restrict__ sbox, u64* __restrict__ gscratch, size_t gscratch_stride, int32_t* __restrict__ ctr)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    __shared__ int wcnt[16], s_n, s_off, s_last;
    const int c = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int32_t* c_in = cls + (size_t)b * N;
    for (int n = tid * C + c; n < N; n += 1024 * C) keep[(size_t)b * N + n] = 0;          // this workgroup's share of the keep flags: n = c (mod C)
    // Each wavefront owns a contiguous range of the candidates (ascending ranges => ascending ids across wavefronts): pass 1 counts its
    // members there, one barrier hands out the wavefronts' start positions, pass 2 compacts with ballots - no barrier inside either pass.
    const int R = (((N + 15) >> 4) + 63) & ~63;             // candidates per wavefront, a multiple of 64
    const int r0 = wave * R, r1 = min(N, r0 + R);
    // (batches of eight labels per lane requested together: one load after the other is a memory round trip per 64 candidates)
    int mine = 0;
    for (int n0 = r0; n0 < r1; n0 += 512) {
        int lab[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) { const int n = n0 + 64 * q + lane; lab[q] = c_in[n < r1 ? n : r1 - 1]; }
#pragma unroll
        for (int q = 0; q < 8; ++q) mine += (n0 + 64 * q + lane < r1 && lab[q] == c) ? 1 : 0;
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) mine += __shfl_xor(mine, o);
    if (lane == 0) wcnt[wave] = mine;
    __syncthreads();
    if (tid == 0) {
        int n = 0;
        for (int w = 0; w < 16; ++w) { const int v = wcnt[w]; wcnt[w] = n; n += v; }      // exclusive prefix: the wavefronts' start positions
        s_n = n;
        s_off = n ? atomicAdd(&ctr[b * 2], n) : 0;
        seg_count[(size_t)b * C + c] = n;
        seg_off[(size_t)b * C + c] = s_off;
    }
    __syncthreads();
    const int n_c = s_n, off = s_off;
    if (n_c) {
        int32_t* ids = bucket + (size_t)b * N + off;
        int pos = wcnt[wave];
        for (int n0 = r0; n0 < r1; n0 += 512) {
            int lab[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) { const int n = n0 + 64 * q + lane; lab[q] = c_in[n < r1 ? n : r1 - 1]; }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int n = n0 + 64 * q + lane;
                const bool hit = n < r1 && lab[q] == c;
                const u64 m = __ballot(hit);
                if (hit) ids[pos + __popcll(m & ((1ull << lane) - 1ull))] = n;
                pos += __popcll(m);
            }
        }
        __threadfence_block();
        __syncthreads();
        const int P = nms_pow2(n_c);
        if (n_c > YN_SORT_LARGE) sort_segment<false>(boxes + (size_t)b * N * 4, scores + (size_t)b * N, ids, true, n_c, P, gscratch + (size_t)b * gscratch_stride, sbox + (size_t)b * N + off);
        else sort_segment<true>(boxes + (size_t)b * N * 4, scores + (size_t)b * N, ids, true, n_c, P, reinterpret_cast<u64*>(sort_lds), sbox + (size_t)b * N + off);
    }
    // the last workgroup of the image: tile offsets (prefix over the classes), counters back to zero
    __threadfence();
    __syncthreads();
    if (tid == 0) s_last = atomicAdd(&ctr[b * 2 + 1], 1) == C - 1 ? 1 : 0;
    __syncthreads();
    if (s_last) {                                           // (block-uniform)
        __threadfence();
        // the C counts with all loads in flight (one agent-scope load after the other is a memory round trip each: 80 x 0.6 us), then the prefix
        int* cnt = reinterpret_cast<int*>(sort_lds);
        for (int cc = tid; cc < C; cc += 1024) cnt[cc] = __hip_atomic_load(&seg_count[(size_t)b * C + cc], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (tid == 0) {
            int tiles = 0;
            for (int cc = 0; cc < C; ++cc) {
                tile_off[(size_t)b * (C + 1) + cc] = tiles;
                const int T = (cnt[cc] + 63) >> 6;
                tiles += T * (T + 1) / 2;
            }
            tile_off[(size_t)b * (C + 1) + C] = tiles;
            ctr[b * 2] = 0; ctr[b * 2 + 1] = 0;
        }
    }
}

// ---- suppression bit-matrix ------------------------------------------------------------------------
// Segment with T = ceil(n/64) chunks owns T(T+1)/2 tiles, stored by bands: band ri holds 64 rows x (T-ri) words,
// word (row, ci-ri) at  band_off(ri) + row*(T-ri) + (ci-ri),  band_off(ri) = 64*(ri*T - ri*(ri-1)/2).
__device__ __forceinline__ size_t band_off(int ri, int T) { return (size_t)64 * ((size_t)ri * T - (size_t)ri * (ri - 1) / 2); }
// Segments ABOVE YN_SORT_SMALL boxes store their tiles column-major instead (round 4): tile (ri, ci) as 64 contiguous words (one per row)
// at ctile_off(ri, ci) - all tiles of a column chunk ci back to back, ri ascending.  What chunk ci's resolve needs from every earlier band is
// then ONE contiguous block (resolve_columns pulls it in with coalesced loads issued four steps ahead), and a tile store is 512 contiguous
// bytes instead of 64 words strided by the band's width.  The same T (T + 1) / 2 * 64 words either way.
__device__ __forceinline__ size_t ctile_off(int ri, int ci) { return ((size_t)ci * (ci + 1) / 2 + ri) * 64; }
__device__ __forceinline__ bool nms_colmajor(int n) { return n > 1024; }

// min / max as plain v_min_f32 / v_max_f32: fminf / fmaxf on operands loaded from LDS cost a canonicalising v_max each (4.5 of the
// 24 VALU instructions per pair in the tile loop below); NaN operands only reach the exact path (union is NaN, not > 0)
__device__ __forceinline__ float vmin_raw(float a, float b) { float r; asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ float vmax_raw(float a, float b) { float r; asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }

// word of sorted item ri*64+lane: bit t = "it suppresses sorted item ci*64+t" (columns after itself only on the diagonal tile)
// core: this lane's row box bx (valid when jr < n) against the chunk's column boxes, this lane's one being cb
template <bool DIOU>
__device__ __forceinline__ u64 tile_word_boxes(const float4 bx, const float4 cb, int n, int ri, int ci, float thresh, float4* cbox, float* carea,
                                               int t_begin = 0, int t_end = 64)      // only columns [t_begin, t_end) of the chunk (multiples of 8)
{
    const int lane = threadIdx.x & 63;
    const float ca = (cb.z - cb.x) * (cb.w - cb.y);
    cbox[lane] = cb;
    carea[lane] = ca;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");             // cbox/carea are private to this wavefront:
    __builtin_amdgcn_wave_barrier();                                   // its LDS accesses execute in order
    const int jr = ri * 64 + lane;
    u64 mask = 0;
    if (jr < n) {
        const float ar = (bx.z - bx.x) * (bx.w - bx.y);
        const int t0 = (ri == ci) ? lane + 1 : 0;
        const int t1 = min(64, n - ci * 64);
        u64 valid = (t1 == 64 ? ~0ull : ((1ull << t1) - 1ull));
        valid &= (t0 >= 64) ? 0ull : ~((1ull << t0) - 1ull);
        if (t_begin > 0 || t_end < 64) valid &= (t_end >= 64 ? ~0ull : ((1ull << t_end) - 1ull)) & ~((1ull << t_begin) - 1ull);
        const int t_stop = t1 < t_end ? t1 : t_end;
        u64 slow = valid;
        if (!DIOU) {
            // Dense, branch-free pass over the chunk's columns (uniform loop, broadcast LDS reads, 8 columns per step, only the
            // t1 live columns): the reference's own arithmetic up to inter and union, then the division-free decision of
            // suppressed() — inter vs thresh*union with a 1e-5 guard band.  Groups of 8 columns holding a pair inside the band (or
            // with union <= 0 / NaN) are re-evaluated by the exact path below (which agrees with the sure decisions by construction).
            u64 sure = 0, unsure = 0;
            // guard band folded into the threshold: hi = fl(u * fl(thr * 1.00001)) >= thr * u * (1 + 0.9e-5), lo likewise below; lo > 1e-30
            // (thr > 0) also says union > 0.  thr <= 0: nothing is decided here, every group goes to the exact path.
            const u64 tmask = thresh > 0.0f ? ~0ull : 0ull;
            const float c_hi = thresh * 1.00001f, c_lo = thresh * 0.99999f;
            for (int t8 = t_begin; t8 < t_stop; t8 += 8) {
                unsigned s8 = 0;                            // bit 7-u = column t8+u surely suppressed (built by shift-in, reversed below)
                unsigned nd = 0;                            // decided pairs of this lane in the group of 8
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const float4 bt = cbox[t8 + u];
                    const float w = vmax_raw(1e-28f, vmin_raw(bx.z, bt.z) - vmax_raw(bx.x, bt.x));
                    const float h = vmax_raw(1e-28f, vmin_raw(bx.w, bt.w) - vmax_raw(bx.y, bt.y));
                    const float inter = w * h;
                    const float un = (ar + carea[t8 + u]) - inter;
                    const float hi = un * c_hi, lo = un * c_lo;
                    const u64 dec = __ballot(lo > 1e-30f) & tmask;
                    const u64 under = __ballot(inter < lo);
                    // over = inter > hi.  s8 = 2 s8 + (over && dec): the compare lands in VCC, the wave masks combine on the scalar unit and
                    // an add-with-carry shifts the bit in; nd += (over || under) && dec the same way — two VALU instructions per pair
                    // where select + or per mask took five.
                    u64 tmp;
                    asm("v_cmp_gt_f32 vcc, %3, %4\n\t"
                        "s_and_b64 %2, vcc, %5\n\t"
                        "s_mov_b64 vcc, %2\n\t"
                        "v_addc_co_u32 %0, vcc, %0, %0, vcc\n\t"
                        "s_and_b64 vcc, %6, %5\n\t"
                        "s_or_b64 vcc, vcc, %2\n\t"
                        "v_addc_co_u32 %1, vcc, 0, %1, vcc"
                        : "+v"(s8), "+v"(nd), "=&s"(tmp) : "v"(inter), "v"(hi), "s"(dec), "s"(under) : "vcc");
                }
                sure |= (u64)(__builtin_bitreverse32(s8) >> 24) << t8;
                unsure |= (nd != 8u) ? (0xffull << t8) : 0ull;          // an undecided pair: the whole group goes to the exact path
            }
            mask = sure & valid;
            slow = unsure & valid;
        }
        while (slow) {
            const int t = __ffsll((long long)slow) - 1;
            slow &= slow - 1;
            const float4 bt = cbox[t];
            if (suppressed(bx, ar, bt, carea[t], thresh, DIOU ? 1 : 0)) mask |= 1ull << t;
        }
    }
    __builtin_amdgcn_wave_barrier();
    return mask;
}

template <bool DIOU>
__device__ __forceinline__ u64 tile_word(const float4* __restrict__ sb, int n, int ri, int ci, float thresh, float4* cbox, float* carea)
{
    const int lane = threadIdx.x & 63;
    const int jc = ci * 64 + lane, jr = ri * 64 + lane;
    float4 cb = make_float4(0.f, 0.f, 0.f, 0.f), bx = cb;
    if (jc < n) cb = sb[jc];
    if (jr < n) bx = sb[jr];
    return tile_word_boxes<DIOU>(bx, cb, n, ri, ci, thresh, cbox, carea);
}

template <bool DIOU, bool COLM = false>
__device__ __forceinline__ void matrix_tile(const float4* __restrict__ sb, int n, int T, int ri, int ci, float thresh,
                                            u64* __restrict__ M, float4* cbox, float* carea)
{
    const u64 mask = tile_word<DIOU>(sb, n, ri, ci, thresh, cbox, carea);
    if (COLM) M[ctile_off(ri, ci) + (threadIdx.x & 63)] = mask;
    else M[band_off(ri, T) + (size_t)(threadIdx.x & 63) * (T - ri) + (ci - ri)] = mask;
}

// grid (G, B), block 64: block g of image b walks tiles g, g+G, ... of that image
// grid (G, B), block 256 = 4 independent wavefronts: wavefront w of block g of image b walks tiles 4g+w, +4G, ...
template <bool DIOU>
__global__ __launch_bounds__(256) void matrix_kernel(const float4* __restrict__ sbox, const int32_t* __restrict__ seg_count,
                                                      const int32_t* __restrict__ seg_off, const int32_t* __restrict__ tile_off,
                                                      int N, int C, float thresh, u64* __restrict__ M, size_t m_stride,
                                                      const int32_t* __restrict__ work_off /* [B][C+1]: tile_off without the segments nms_sweep_kernel takes, or null (= tile_off) */,
                                                      const int32_t* __restrict__ seg_sparse, const int32_t* __restrict__ large_list, int large_cap)
{
    __shared__ float4 cbox_all[4][64];
    __shared__ float carea_all[4][64];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));      // wave-uniform for the compiler too: tile indices, masks and ballots stay in SGPRs
    float4* cbox = cbox_all[wave];
    float* carea = carea_all[wave];
    const int b = blockIdx.y;
    const int32_t* soff = tile_off + (size_t)b * (C + 1);                 // storage: where a segment's tiles live
    const int32_t* toff = (work_off ? work_off : tile_off) + (size_t)b * (C + 1);   // work: the tiles this kernel evaluates (a marked segment has none)
    const int waves = (int)gridDim.x * 4, wid = (int)blockIdx.x * 4 + wave;
    // the marked segments' tiles: zeros, spread over the image's wavefronts (nms_sweep_kernel ORs its few hits into them afterwards)
    if (seg_sparse && large_list) {
        const int32_t* ll = large_list + (size_t)b * (large_cap + 1);
        const int nl = ll[0];
        for (int k = 0; k < nl; ++k) {
            const int c = ll[1 + k];
            if (!seg_sparse[(size_t)b * C + c]) continue;
            u64* Ms = M + (size_t)b * m_stride + (size_t)soff[c] * 64;
            const int nt = soff[c + 1] - soff[c];
            for (int t = wid; t < nt; t += waves) Ms[(size_t)t * 64 + (threadIdx.x & 63)] = 0ull;
        }
    }
    const int total = toff[C];
    // A wavefront walks a CONTIGUOUS range of the image's tiles (round 5; it took tiles 4g + w, + 4G, ...): locating a tile - a binary search over
    // the classes' tile offsets, the (row chunk, column chunk) of the triangle by subtraction, the segment's count and offset: a dozen dependent
    // scalar loads, ~2.5 k cycles beside the tile's ~5 k - is paid once per range; the next tile is one step along the row (or the start of
    // the next row / segment).  Same tiles, same words.
    const int per = (total + waves - 1) / waves;
    int t = wid * per;
    const int t_end = min(total, t + per);
    if (t >= t_end) return;
    int lo = 0, hi = C;                                     // largest c with toff[c] <= t
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (toff[mid] <= t) lo = mid; else hi = mid; }
    int c = lo;
    while (toff[c + 1] <= t) ++c;                           // (segments without work share their successor's offset: the one with tiles is the last of a run of equal offsets)
    int n = seg_count[(size_t)b * C + c];
    int T = (n + 63) >> 6;
    int rem = t - toff[c], ri = 0;
    while (rem >= T - ri) { rem -= T - ri; ++ri; }
    int ci = ri + rem;
    for (;;) {
        const float4* sb = sbox + (size_t)b * N + seg_off[(size_t)b * C + c];
        u64* Ms = M + (size_t)b * m_stride + (size_t)soff[c] * 64;
        if (nms_colmajor(n)) matrix_tile<DIOU, true>(sb, n, T, ri, ci, thresh, Ms, cbox, carea);
        else matrix_tile<DIOU>(sb, n, T, ri, ci, thresh, Ms, cbox, carea);
        if (++t >= t_end) break;
        if (++ci == T) {
            if (++ri == T) {                                // the segment's last tile: on to the next class that has any
                do { ++c; } while (toff[c + 1] <= t);
                n = seg_count[(size_t)b * C + c];
                T = (n + 63) >> 6;
                ri = 0;
            }
            ci = ri;
        }
    }
}

// ---- resolve ---------------------------------------------------------------------------------------
// One workgroup (256 threads) per segment.  rem[w] (LDS) = removed mask of chunk w.  Per chunk: wave 0
// resolves the diagonal tile serially (scalar readlanes; the next diagonal is prefetched meanwhile), then
// all threads OR the kept rows' words of the band into rem[] (independent loads, LDS atomics on the few
// non-zero words).  Returns the number of kept boxes.
// rem[] bounds the segment size: n <= 64 * YN_RESOLVE_MAX_T = 131 072 boxes of one class (nms_max_segment(); the C ABI
// rejects larger work before anything is launched — the suppression matrix of such a segment would be > 1 GB anyway)
#define YN_RESOLVE_MAX_T 2048
struct ResolveLds { u64 rem[YN_RESOLVE_MAX_T]; int pbase[YN_RESOLVE_MAX_T]; u64 keepm; u64 keepm2[2]; int kidx[64]; int nk; };     // keepm2: band ri's kept mask in slot ri & 1 (resolve_bands)

// Everything a band reads from memory — its diagonal word and candidate id (wave 0), its off-diagonal words (all threads:
// thread -> row tid/4, columns 1 + (tid&3) + 4u) — is requested TWO bands ahead, into one of three register sets: a band's own work
// is a few hundred cycles, a load of the freshly written matrix ~3 000 (it misses L2), and with a look-ahead of one band (the first
// version) every band waited for its loads: 132 k cycles for the 39 bands of the benchmark's largest segment, 37 serial steps in all
// (81 -> 66 us; issuing the loads unconditionally at clamped addresses to keep the vmcnt bookkeeping static was slower: 104 us).
template <int NB>
struct ResolvePre { u64 nb[NB]; u64 diag, c1; int id; };       // c1 (wave 0): the band's column-1 word of row `lane`

template <int NB, int TPR>
__device__ __forceinline__ void resolve_prefetch(ResolvePre<NB>& p, const int32_t* __restrict__ ids, int n, const u64* __restrict__ M, int T, int ri,
                                                 int lane, int wave, int pr, int pc)
{
    if (ri >= T) return;
    const int W = T - ri;
    if (wave == 0) {
        p.diag = M[band_off(ri, T) + (size_t)lane * W];
        p.c1 = W > 1 ? M[band_off(ri, T) + (size_t)lane * W + 1] : 0ull;
    }
    // no per-word branches (each was a saveexec / branch region: ~40 of them per band made the walk instruction-bound at one or two
    // wavefronts per SIMD): words past the row's end are read at a clamped column and ignored where they would be used
    const size_t boff = band_off(ri, T) + (size_t)pr * W;
#pragma unroll
    for (int u = 0; u < NB; ++u) {
        const int w = 1 + pc + TPR * u;
        p.nb[u] = M[boff + (w < W ? w : W - 1)];
    }
}

// wave 0: which boxes of chunk ri survive (-> L.keepm, L.kidx, L.nk; keep flags / pick list)
// DEFER (resolve_bands): nothing goes to global memory here — on this architecture stores count in vmcnt like loads, so a keep-flag store
// per band made every later wait for prefetched matrix words wait for that store's acknowledgement as well: ~1.8 us per band whatever
// the look-ahead.  The kept mask replaces rem[ri] (dead once the band is resolved) and the flags / pick list are written after the walk.
template <bool DEFER = false>
__device__ __forceinline__ u64 resolve_diag(u64 diag, int id, int n, int ri, int lane, int picked,
                                            int32_t* __restrict__ keep_flags, int32_t* __restrict__ pick_list, ResolveLds& L)
{
    const int cnt = min(64, n - ri * 64);
    const u64 validm = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
    const u64 alive0 = validm & ~L.rem[ri];
    u64 keepm = alive0;
    // only the alive rows that suppress an alive column take a serial step (none in the common case): a row's word only holds
    // later columns, so whatever is still alive once those rows are through is kept
    u64 work = __ballot(((alive0 >> lane) & 1ull) && (diag & alive0));
    if (work != 0ull) {
        const unsigned dlo = (unsigned)(diag & 0xffffffffu), dhi = (unsigned)(diag >> 32);
        u64 alive = alive0;
        while (work) {
            const int i = __ffsll((long long)work) - 1;                                  // alive at its turn: kept
            const unsigned lo_i = (unsigned)__builtin_amdgcn_readlane((int)dlo, i);     // (unsigned): no sign extension
            const unsigned hi_i = (unsigned)__builtin_amdgcn_readlane((int)dhi, i);
            alive &= ~(((u64)hi_i << 32) | (u64)lo_i);
            work &= alive & ~(1ull << i);                                                // bits <= i of later words are zero
        }
        keepm = alive;
    }
    if (DEFER) {
        if (lane == 0) { L.rem[ri] = keepm; L.pbase[ri] = picked; L.keepm2[ri & 1] = keepm; }
        return keepm;
    }
    if ((keepm >> lane) & 1ull) {
        const int rank = __popcll(keepm & ((1ull << lane) - 1ull));
        L.kidx[rank] = lane;
        if (keep_flags) keep_flags[id] = 1;
        if (pick_list) pick_list[picked + rank] = id;
    }
    if (lane == 0) { L.nk = __popcll(keepm); L.keepm = keepm; L.keepm2[ri & 1] = keepm; }
    return keepm;
}

template <int NB, int TPR = 4>                               // TPR threads per matrix row (64 * TPR threads per workgroup), NB words per thread
__device__ __forceinline__ int resolve_bands(const int32_t* __restrict__ ids, int n, const u64* __restrict__ M,
                                             int32_t* __restrict__ keep_flags, int32_t* __restrict__ pick_list, ResolveLds& L)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int T = (n + 63) >> 6;
    const int pr = tid / TPR, pc = tid % TPR;
    ResolvePre<NB> P0, P1, P2;
    resolve_prefetch<NB, TPR>(P0, ids, n, M, T, 0, lane, wave, pr, pc);
    resolve_prefetch<NB, TPR>(P1, ids, n, M, T, 1, lane, wave, pr, pc);
    __syncthreads();
    // One barrier per band, the serial step overlapped with the bulk work: in iteration ri wavefront 0 resolves the diagonal tile of band
    // ri and ORs the kept rows' COLUMN-1 words into rem[ri + 1] itself (its own LDS operations execute in order, so it sees them when it
    // resolves band ri + 1), while all threads apply band ri - 1's remaining words — kept mask from slot (ri - 1) & 1 — to rem[ri + 1 ...]:
    // what diag(ri + 1) needs from bands <= ri - 1 is complete at the barrier that ends iteration ri.  (First version: diag, barrier,
    // apply, barrier per band, 1.9 us each: 159 us for the 83 bands of the largest class at 608 x 608, bs 1.)
    int picked = 0;
#define YN_BAND(P, PPREV, ri_)                                                                               \
    {                                                                                                        \
        const int ri = (ri_);                                                                                \
        if (wave == 0) {                                                                                     \
            const u64 km = resolve_diag<true>(P.diag, 0, n, ri, lane, picked, keep_flags, pick_list, L);     \
            picked += __popcll(km);                                                                          \
            if (((km >> lane) & 1ull) && P.c1) atomicOr(&L.rem[ri + 1], P.c1);                               \
        }                                                                                                    \
        if (ri >= 1 && ((L.keepm2[(ri - 1) & 1] >> pr) & 1ull)) {                                            \
            u64 any = 0;                                                                                     \
            _Pragma("unroll") for (int u = 0; u < NB; ++u) any |= PPREV.nb[u];                               \
            if (any) {                                       /* rare: most kept rows suppress nothing */     \
                _Pragma("unroll") for (int u = 0; u < NB; ++u)                                               \
                    if (pc + TPR * u > 0 && ri + pc + TPR * u < T && PPREV.nb[u]) atomicOr(&L.rem[ri + pc + TPR * u], PPREV.nb[u]); /* column 1 is wavefront 0's */ \
            }                                                                                                \
        }                                                                                                    \
        resolve_prefetch<NB, TPR>(PPREV, ids, n, M, T, ri + 2, lane, wave, pr, pc);                          \
        __syncthreads();                                                                                     \
    }
    for (int r0 = 0; r0 < T; r0 += 3) {
        YN_BAND(P0, P2, r0)
        if (r0 + 1 < T) YN_BAND(P1, P0, r0 + 1)
        if (r0 + 2 < T) YN_BAND(P2, P1, r0 + 2)
    }
#undef YN_BAND
    if (tid == 0) L.nk = picked;
    __syncthreads();
    // rem[ri] now holds band ri's kept mask: the keep flags / the pick list (np order: score-descending inside the class) in one coalesced pass
    for (int i = tid; i < n; i += blockDim.x) {
        const int ri = i >> 6, l = i & 63;
        const u64 km = L.rem[ri];
        if ((km >> l) & 1ull) {
            const int id = ids[i];
            if (keep_flags) keep_flags[id] = 1;
            if (pick_list) pick_list[L.pbase[ri] + __popcll(km & ((1ull << l) - 1ull))] = id;
        }
    }
    return L.nk;
}

// ---- the walk of a LARGE segment (column-major tiles): what chunk c needs is pulled, not pushed ------------------------------------------
// resolve_bands pushes band ri's words forward into the masks of later chunks and reads them where the matrix kernel left them: 64 rows x
// (T - ri) words strided by the band's width, requested two bands ahead - 1.4 us per band, 116 us for the 79 bands of the 5 000-box class
// of one 608 x 608 image (its latency is the loaded words', not the 37 serial steps').  Here chunk c PULLS: removed(c) = OR over the kept
// rows of every earlier band of that band's word for c - with column-major tiles ONE contiguous block of 64 c words, requested FOUR steps
// ahead with coalesced loads (wavefronts 1-7: a thread's words of a block sit in four rotating register sets), masked with the earlier
// chunks' kept masks (complete by then) and OR-ed into rem[c + 1] one step before wavefront 0 needs it; wavefront 0 only walks the
// diagonal tiles (resolve_diag) and adds the one tile that depends on the chunk it has just resolved (band c -> chunk c + 1).
// One barrier per step, no load on the serial path.  512 threads; NB words per thread and block: T <= 7 NB + 1 chunks.
// Measured (one stream, HIP-event brackets, resolve launches of a step): 608 x 608 one image 116 -> 104 us, 416 x 416 one image 45 -> 34,
// 608 x 608 bs 32 115 -> 102, 416 x 416 bs 32 45 -> 35.
// OR of a 64-bit value over the wavefront (wave-uniform result): rotations inside the rows of 16 lanes (DPP row_ror 8 / 4 / 2 / 1), then the
// four rows through readlane.  ~30 instructions; 64 lanes' atomicOr on ONE LDS word serialise to over a thousand cycles.
__device__ __forceinline__ unsigned row_or16(unsigned v)
{
    v |= (unsigned)dpp_i<0x128>((int)v); v |= (unsigned)dpp_i<0x124>((int)v); v |= (unsigned)dpp_i<0x122>((int)v); v |= (unsigned)dpp_i<0x121>((int)v);
    return v;
}
__device__ __forceinline__ u64 wave_or64(u64 v)
{
    const unsigned lo = row_or16((unsigned)(v & 0xffffffffu)), hi = row_or16((unsigned)(v >> 32));
    const unsigned l = (unsigned)__builtin_amdgcn_readlane((int)lo, 0) | (unsigned)__builtin_amdgcn_readlane((int)lo, 16) |
                       (unsigned)__builtin_amdgcn_readlane((int)lo, 32) | (unsigned)__builtin_amdgcn_readlane((int)lo, 48);
    const unsigned h = (unsigned)__builtin_amdgcn_readlane((int)hi, 0) | (unsigned)__builtin_amdgcn_readlane((int)hi, 16) |
                       (unsigned)__builtin_amdgcn_readlane((int)hi, 32) | (unsigned)__builtin_amdgcn_readlane((int)hi, 48);
    return ((u64)h << 32) | (u64)l;
}

template <int NB>
struct ColPre { ulonglong2 nb[NB / 2]; u64 diag, prev; };     // a thread's words of a block: NB / 2 pairs (16-byte loads)

template <int NB>
__device__ __forceinline__ void column_prefetch(ColPre<NB>& p, const u64* __restrict__ M, int T, int cc, int lane, int q)
{
    // EVERY thread issues the same NB / 2 + 2 loads, at clamped addresses, outside any branch: wavefront 0 does not use its block words, the
    // others not their diagonal words, steps past the walk's end re-read the last column - the compiler can then count the loads in flight
    // (s_waitcnt vmcnt(36 .. 51) where a step's words are used; with the loads inside `if (wave == 0) ... else ...` / `if (cc < T)` every
    // join was a vmcnt(0)).  Measured on the 5 400-box class of one 608 x 608 image, in the order tried: first form 187 us; counted waits
    // 187 (the wait was not the bound); 16-byte loads 133; one atomic per wavefront instead of one per lane (wave_or64) 104 - against 116
    // for resolve_bands.  What is left is the block traffic itself: 1.85 MB of freshly written tiles through ONE CU.
    cc = cc < T ? cc : T - 1;
    p.diag = M[ctile_off(cc, cc) + lane];
    p.prev = M[ctile_off(cc >= 1 ? cc - 1 : 0, cc) + lane];
    // (16-byte loads, two words per lane; pairs past the block's end all read pair 0: one cache line per wavefront)
    const int pairs = 32 * (cc - 1);                        // tiles ri <= cc - 2 of column cc: contiguous, 32 pairs each
    const ulonglong2* blk = reinterpret_cast<const ulonglong2*>(M + ctile_off(0, cc));
#pragma unroll
    for (int u = 0; u < NB / 2; ++u) {
        const int idx = q + 448 * u;
        p.nb[u] = blk[(idx >= 0 && idx < pairs) ? idx : 0];
    }
}

template <int NB>
__device__ __forceinline__ int resolve_columns(const int32_t* __restrict__ ids, int n, const u64* __restrict__ M,
                                               int32_t* __restrict__ keep_flags, int32_t* __restrict__ pick_list, ResolveLds& L)
{
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));   // wave-uniform for the compiler too: the role branches below are scalar
    const int T = (n + 63) >> 6;
    const int q = tid - 64;                                 // wavefronts 1-7: 448 threads share a block's words
    ColPre<NB> P0, P1, P2, P3;
    column_prefetch<NB>(P0, M, T, 0, lane, q);
    column_prefetch<NB>(P1, M, T, 1, lane, q);
    column_prefetch<NB>(P2, M, T, 2, lane, q);
    column_prefetch<NB>(P3, M, T, 3, lane, q);
    __syncthreads();
    int picked = 0;
#ifdef YN_EXP_TIMING
    long long t_work = 0, t_pre = 0, t_bar = 0, t_all0 = __builtin_readcyclecounter();
#define YN_T0() long long tt0 = __builtin_readcyclecounter()
#define YN_T1() long long tt1 = __builtin_readcyclecounter()
#define YN_T2() long long tt2 = __builtin_readcyclecounter()
#define YN_T3() { long long tt3 = __builtin_readcyclecounter(); t_work += tt1 - tt0; t_pre += tt2 - tt1; t_bar += tt3 - tt2; }
#else
#define YN_T0()
#define YN_T1()
#define YN_T2()
#define YN_T3()
#endif
    // step c: wavefront 0 resolves chunk c (set P) and pushes tile (c, c + 1) (set PN.prev) into rem[c + 1]; the others apply the block of
    // column c + 1 (set PN.nb: tiles ri <= c - 1, whose kept masks stand in rem[ri] since the barrier that ended step c - 1); P is then
    // refilled with column c + 4.  The walk runs in whole rounds of four steps: steps past the last chunk only load.
#define YN_COL(P, PN, c_)                                                                                    \
    {                                                                                                        \
        const int c = (c_);                                                                                  \
        YN_T0();                                                                                             \
        if (c < T) {                                                                                         \
            if (wave == 0) {                                                                                 \
                const u64 km = resolve_diag<true>(P.diag, 0, n, c, lane, picked, keep_flags, pick_list, L);  \
                picked += __popcll(km);                                                                      \
                if (c + 1 < T) {                             /* tile (c, c + 1) of the kept rows: one atomic, not one per lane */ \
                    const u64 pv = ((km >> lane) & 1ull) ? PN.prev : 0ull;                                   \
                    if (__ballot(pv != 0ull)) { const u64 r = wave_or64(pv); if (lane == 0) atomicOr(&L.rem[c + 1], r); } \
                }                                                                                            \
            } else if (c >= 1 && c + 1 < T) {                                                                \
                const int pairs = 32 * c;                                                                    \
                u64 any = 0;                                                                                 \
                _Pragma("unroll") for (int u = 0; u < NB / 2; ++u) any |= (q + 448 * u < pairs) ? (PN.nb[u].x | PN.nb[u].y) : 0ull; \
                if (__ballot(any != 0ull)) {                 /* (wave-uniform) most rows suppress nothing */  \
                    u64 v = 0;                                                                               \
                    _Pragma("unroll") for (int u = 0; u < NB / 2; ++u) {                                     \
                        const int idx = 2 * (q + 448 * u);   /* word index of .x: row idx & 63 of tile idx >> 6 */ \
                        if (idx < 2 * pairs) {                                                               \
                            const u64 km = L.rem[idx >> 6];                                                  \
                            if ((km >> (idx & 63)) & 1ull) v |= PN.nb[u].x;                                  \
                            if ((km >> ((idx & 63) + 1)) & 1ull) v |= PN.nb[u].y;                            \
                        }                                                                                    \
                    }                                                                                        \
                    if (__ballot(v != 0ull)) { const u64 r = wave_or64(v); if (lane == 0) atomicOr(&L.rem[c + 1], r); } \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
        YN_T1();                                                                                             \
        column_prefetch<NB>(P, M, T, c + 4, lane, q);                                                        \
        YN_T2();                                                                                             \
        __syncthreads();                                                                                     \
        YN_T3();                                                                                             \
    }
    for (int c0 = 0; c0 < T; c0 += 4) {
        YN_COL(P0, P1, c0)
        YN_COL(P1, P2, c0 + 1)
        YN_COL(P2, P3, c0 + 2)
        YN_COL(P3, P0, c0 + 3)
    }
#undef YN_COL
#ifdef YN_EXP_TIMING
    if ((tid == 0 || tid == 64) && T > 30)
        printf("resolvecol n %d T %d wave %d work %lld prefetch %lld barrier %lld total %lld per step %lld\n", n, T, wave, t_work, t_pre, t_bar,
               (long long)__builtin_readcyclecounter() - t_all0, ((long long)__builtin_readcyclecounter() - t_all0) / T);
#endif
    if (tid == 0) L.nk = picked;
    __syncthreads();
    // rem[c] now holds chunk c's kept mask: the keep flags / the pick list in one coalesced pass
    for (int i = tid; i < n; i += blockDim.x) {
        const int ri = i >> 6, l = i & 63;
        const u64 km = L.rem[ri];
        if ((km >> l) & 1ull) {
            const int id = ids[i];
            if (keep_flags) keep_flags[id] = 1;
            if (pick_list) pick_list[L.pbase[ri] + __popcll(km & ((1ull << l) - 1ull))] = id;
        }
    }
    return L.nk;
}

// any T: the same pull without register staging (segments above 7 232 boxes: chunk c's block is read when it is needed)
__device__ __forceinline__ int resolve_columns_unstaged(const int32_t* __restrict__ ids, int n, const u64* __restrict__ M,
                                                        int32_t* __restrict__ keep_flags, int32_t* __restrict__ pick_list, ResolveLds& L)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int T = (n + 63) >> 6;
    __syncthreads();
    int picked = 0;
    for (int c = 0; c < T; ++c) {
        const u64* blk = M + ctile_off(0, c);
        for (int idx = tid; idx < 64 * c; idx += nthr) {
            const u64 w = blk[idx];
            if (w && ((L.rem[idx >> 6] >> (idx & 63)) & 1ull)) atomicOr(&L.rem[c], w);
        }
        __syncthreads();
        if (wave == 0) picked += __popcll(resolve_diag<true>(M[ctile_off(c, c) + lane], 0, n, c, lane, picked, keep_flags, pick_list, L));
        __syncthreads();
    }
    if (tid == 0) L.nk = picked;
    __syncthreads();
    for (int i = tid; i < n; i += nthr) {
        const int ri = i >> 6, l = i & 63;
        const u64 km = L.rem[ri];
        if ((km >> l) & 1ull) {
            const int id = ids[i];
            if (keep_flags) keep_flags[id] = 1;
            if (pick_list) pick_list[L.pbase[ri] + __popcll(km & ((1ull << l) - 1ull))] = id;
        }
    }
    return L.nk;
}

// MODE selects the staged walks a kernel instantiates (its register count is the largest one's): 0 = 256 threads, T <= 17 only
// (resolve_kernel: segments up to 1 024 boxes); 1 = 512 threads, eight per matrix row, up to 129 chunks (resolve_large_kernel); 2 = 256
// threads, up to 65 chunks (single_resolve_kernel).  Anything larger takes the unstaged loop below.
template <int MODE>
__device__ __forceinline__ int resolve_segment(const int32_t* __restrict__ ids, int n, const u64* __restrict__ M,
                                               int32_t* __restrict__ keep_flags, int32_t* __restrict__ pick_list, ResolveLds& L)
{
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int T = (n + 63) >> 6;
    for (int w = tid; w < T; w += nthr) L.rem[w] = 0;
    if (MODE == 0 && nthr == 256) {
        if (T <= 17) return resolve_bands<4>(ids, n, M, keep_flags, pick_list, L);
    }
    if (MODE == 2 && nthr == 256) {                         // the band words of a row fit four threads' register sets
        if (T <= 17) return resolve_bands<4>(ids, n, M, keep_flags, pick_list, L);
        if (T <= 33) return resolve_bands<8>(ids, n, M, keep_flags, pick_list, L);
        if (T <= 65) return resolve_bands<16>(ids, n, M, keep_flags, pick_list, L);
    }
    if (MODE == 1 && nthr == 512) {
        if (nms_colmajor(n)) {                              // above YN_SORT_SMALL boxes: column-major tiles, the pulling walk
            if (T <= 29) return resolve_columns<4>(ids, n, M, keep_flags, pick_list, L);
            if (T <= 57) return resolve_columns<8>(ids, n, M, keep_flags, pick_list, L);
            if (T <= 113) return resolve_columns<16>(ids, n, M, keep_flags, pick_list, L);
            return resolve_columns_unstaged(ids, n, M, keep_flags, pick_list, L);
        }
        return resolve_bands<4, 8>(ids, n, M, keep_flags, pick_list, L);       // (few-segment batches: the small segments ride along; T <= 17)
    }
    // very large segments (n > 4160): no register staging, the kept rows' words are read when they are needed
    u64 diag_next = 0;
    int id_next = 0;
    if (wave == 0) {
        diag_next = M[(size_t)lane * T];
        id_next = lane < n ? ids[lane] : 0;
    }
    __syncthreads();
    int picked = 0;
    for (int ri = 0; ri < T; ++ri) {
        const size_t boff = band_off(ri, T);
        const int W = T - ri;
        if (wave == 0) {
            const u64 diag = diag_next;
            const int id = id_next;
            if (ri + 1 < T) {
                diag_next = M[band_off(ri + 1, T) + (size_t)lane * (W - 1)];
                id_next = ((ri + 1) * 64 + lane < n) ? ids[(ri + 1) * 64 + lane] : 0;
            }
            resolve_diag(diag, id, n, ri, lane, picked, keep_flags, pick_list, L);
        }
        __syncthreads();
        const int nk = L.nk;
        picked += nk;
        const int Wr = W - 1;                               // later chunks of this band
        const int total = nk * Wr;
        for (int q = tid; q < total; q += nthr) {
            const int k = q / Wr, w = 1 + (q - k * Wr);
            const u64 v = M[boff + (size_t)L.kidx[k] * W + w];
            if (v) atomicOr(&L.rem[ri + w], v);
        }
        __syncthreads();
    }
    return picked;
}

__global__ __launch_bounds__(256) void resolve_kernel(const int32_t* __restrict__ seg_count, const int32_t* __restrict__ seg_off,
                                                       const int32_t* __restrict__ tile_off, const int32_t* __restrict__ bucket,
                                                       int N, int C, const u64* __restrict__ M, size_t m_stride, int32_t* __restrict__ keep, int n_max,
                                                       const int32_t* __restrict__ seg_order)
{
    __shared__ ResolveLds L;
    const int b = blockIdx.x;                               // grid (B, C): largest segments of every image first
    const int c = seg_order ? seg_order[(size_t)b * C + blockIdx.y] : (int)blockIdx.y;
    const int n = seg_count[(size_t)b * C + c];
    if (n == 0 || n > n_max) return;                        // n > n_max: resolve_large_kernel's
    resolve_segment<0>(bucket + (size_t)b * N + seg_off[(size_t)b * C + c], n,
                    M + (size_t)b * m_stride + (size_t)tile_off[(size_t)b * (C + 1) + c] * 64, keep + (size_t)b * N, nullptr, L);
}

// The segments of bucket_kernel's large list (more than 1 024 boxes before the prefilter) that still hold more than YN_SORT_SMALL boxes:
// ONE launch of 512-thread workgroups, eight threads per matrix row, so that the few long walks of an image run side by side (a
// 256-thread walk up to 65 chunks followed by a second kernel for the larger ones ran the two longest walks of a 608 x 608 image one
// after the other: 155 us for 65 + 83 bands) and resolve_kernel — 2 560 mostly tiny segments at bs 32 — only carries the registers of
// the shortest staged walk.
__global__ __launch_bounds__(512) void resolve_large_kernel(const int32_t* __restrict__ seg_count, const int32_t* __restrict__ seg_off,
                                                             const int32_t* __restrict__ tile_off, const int32_t* __restrict__ bucket,
                                                             int N, int C, const u64* __restrict__ M, size_t m_stride, int32_t* __restrict__ keep,
                                                             const int32_t* __restrict__ large_list, int large_cap, int n_min)
{
    __shared__ ResolveLds L;
    // large_list == null: grid (C, B), every segment (few segments: one launch for all of them).  With the list: grid (B, slots) - the image fastest:
    // workgroups go to the XCDs round robin by linear id, and a slot count that is a multiple of eight in x would put an image's few long walks
    // (slots 0 - 2) on the same three XCDs for every image (nms_sweep_kernel's grid did exactly that, round 6)
    const int b = large_list ? blockIdx.x : blockIdx.y;
    int c = large_list ? blockIdx.y : blockIdx.x;
    if (large_list) {
        const int32_t* ll = large_list + (size_t)b * (large_cap + 1);
        if (c >= ll[0]) return;
        c = ll[1 + c];
    }
    const int n = seg_count[(size_t)b * C + c];
    if (n <= n_min) return;
    resolve_segment<1>(bucket + (size_t)b * N + seg_off[(size_t)b * C + c], n,
                       M + (size_t)b * m_stride + (size_t)tile_off[(size_t)b * (C + 1) + c] * 64, keep + (size_t)b * N, nullptr, L);
}

// ---- first-chunk prefilter ---------------------------------------------------------------------------------------------------------
// The bit matrix costs n^2/2 pairs per class, and on clustered classes almost all of them are wasted: the benchmark's 1 181-box class
// keeps 13 boxes.  The 64 best-scored boxes of a segment can be resolved on their own (nothing outside the chunk precedes them), and
// every later box one of their KEPT boxes suppresses is out for good — a removed box suppresses nothing, so dropping it before the dense
// phase cannot change any other decision.  One workgroup per segment: diagonal tile of chunk 0 + its serial resolve (wave 0), then the
// band (0, ci) for every later chunk (the dense tile code, row words OR-ed over the kept rows), then an order-preserving compaction of
// the survivors into a second candidate list.  matrix / resolve run on that list (chunk 0 is finished: its kept boxes are flagged here).
// Segments of <= 64 boxes never reach the dense phase.  Exact: the kept sets equal the plain pipeline's (parity suite).
// (nms_sweep_kernel's constants and box classes: its decision is taken at the end of the prefilter, the kernel itself follows below)
#define YN_SWEEP_BINS 1024
#define YN_SWEEP_MAXN 6144                                 // boxes (16 B) + the bin-sorted index (4 B) in LDS: 120 KB
#define YN_SWEEP_IRR 64
#define YN_SWEEP_WIDE 256                                  // listed wide boxes per workgroup (more: sixteen lanes each, as the narrow ones)
#define YN_SWEEP_WIDE_VISITS 256                           // a box with more visits than this is wide
// (workgroups per segment in the pairs phase - each bins the segment itself, then takes every KZ-th box: chosen at launch, launch_nms_pipeline)
struct SweepBins {                                          // the bin of an x coordinate: monotone in x (one rounding per step, each monotone), clamped
    float xlo, scale;
    __device__ __forceinline__ SweepBins(unsigned lo_u, unsigned hi_u)      // the x-range as ordered bits (order_bits)
    {
        xlo = __uint_as_float((lo_u & 0x80000000u) ? (lo_u & 0x7fffffffu) : ~lo_u);
        const float xhi = __uint_as_float((hi_u & 0x80000000u) ? (hi_u & 0x7fffffffu) : ~hi_u);
        scale = (xhi > xlo) ? (float)YN_SWEEP_BINS / (xhi - xlo) : 0.0f;
    }
    __device__ __forceinline__ int of(float x) const
    {
        const float f = (x - xlo) * scale;
        int q = (int)f;
        q = f >= (float)YN_SWEEP_BINS ? YN_SWEEP_BINS - 1 : q;
        return q < 0 ? 0 : q;
    }
};
__device__ __forceinline__ bool sweep_regular(const float4 b)
{
    const float w = b.z - b.x, a = w * (b.w - b.y);
    return (b.x - b.x == 0.0f) && (b.y - b.y == 0.0f) && (b.z - b.z == 0.0f) && (b.w - b.w == 0.0f) && w > 1e-20f && a >= 1e-20f && a < 3.0e38f;
}
#define YN_PRE_W 8                                         // wavefronts per prefilter workgroup
// A LARGE segment (more than 1 024 boxes) was one workgroup's work for 40-60 us - 38-85 band tiles on half a CU, ~5 us per round of eight, while
// the other 2 500 workgroups of the launch are done after 25 us (measured with s_memrealtime stamps, round 6).  Its band is SLICED over up to
// YN_PRE_Z workgroups instead: each resolves chunk 0 for itself (the same tile, the same serial walk: identical kept masks), takes every ZS-th
// round of band tiles and leaves its survivor words in global memory (write-through stores, one ticket per workgroup); the workgroup that
// draws the last ticket reads all words back and compacts.  No workgroup waits for another.  Sliced: the first YN_PRE_RANKS segments of an
// image in seg_order (by size - the only ones that can be large in practice; a fifth one stays on one workgroup), grid.y = C + RANKS * YN_PRE_Z
// with the slices first.  One MORE workgroup per sliced segment takes the sweep's decision (below) beside the band slices.
#define YN_PRE_Z 8
#define YN_PRE_RANKS 4
#define YN_PRE_TICKETS 4096                                // pre_sync: [YN_PRE_TICKETS] ticket counters (image, rank) - a fixed region, zero between launches whatever B and N were -
__host__ __device__ inline size_t pre_sync_words(int N) { return (size_t)(N / 64 + 2); }             // then per (image, rank): word 0 = the sweep's decision, words 1 .. T - 1 = a survivor word per chunk
struct PrefilterLds { u64 surv[YN_RESOLVE_MAX_T]; int base[YN_RESOLVE_MAX_T]; u64 keepm; int last; __attribute__((aligned(8))) unsigned char dpart[64][YN_PRE_W]; int part[64 * YN_PRE_W]; float4 cbox[YN_PRE_W][64]; float carea[YN_PRE_W][64]; };

// The sweep's DECISION for a large segment (a launch of its own at first - 22 us of 1 024-thread workgroups -, then the tail of the compacting
// workgroup - 16 us on the launch's critical path; now a workgroup of its own beside the band slices): would nms_sweep_kernel visit at most a
// quarter of the n^2 / 2 pairs?  Bins of the left edges counted in LDS (the compaction's arrays), a visit count from their prefix sums; not more
// than YN_SWEEP_IRR irregular boxes - the classifier and the bin map are nms_sweep_kernel's own.  Taken on the segment's boxes BEFORE the
// prefilter (chunks >= 1): the survivors are a subset - fewer irregular boxes, fewer visits -, and either answer is correct (it only chooses
// between two exact ways to the same suppression words).  Every thread of the workgroup calls it; block-uniform result.
__device__ bool sweep_spread_out(const float4* __restrict__ sb, int n, PrefilterLds& L)
{
    constexpr int NTH = 64 * YN_PRE_W;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int* hist = L.base;                                                  // [YN_SWEEP_BINS + 1]
    unsigned* sc = reinterpret_cast<unsigned*>(L.part);                  // [0] min left edge, [1] max right edge (ordered bits), [2] irregular boxes
    for (int i = tid; i < YN_SWEEP_BINS + 1; i += NTH) hist[i] = 0;
    if (tid == 0) { sc[0] = 0xffffffffu; sc[1] = 0u; sc[2] = 0u; L.keepm = 0ull; }
    __syncthreads();
    constexpr int PER = (YN_SWEEP_MAXN + NTH - 1) / NTH;                 // boxes per thread: up to 12 at 6 144
    if (n - 64 > PER * NTH) return false;
    {
        unsigned mn = 0xffffffffu, mx = 0u, irr = 0u;
        for (int i = 64 + tid; i < n; i += NTH) {
            const float4 bx = sb[i];
            if (sweep_regular(bx)) { mn = min(mn, order_bits(bx.x)); mx = max(mx, order_bits(bx.z)); }
            else ++irr;
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { mn = min(mn, (unsigned)__shfl_xor((int)mn, o)); mx = max(mx, (unsigned)__shfl_xor((int)mx, o)); irr += (unsigned)__shfl_xor((int)irr, o); }
        if (lane == 0) { atomicMin(&sc[0], mn); atomicMax(&sc[1], mx); if (irr) atomicAdd(&sc[2], irr); }
    }
    __syncthreads();
    const int nirr = (int)sc[2];
    if (nirr > YN_SWEEP_IRR) return false;
    const SweepBins bins(sc[0], sc[1]);
    unsigned qq[PER];                                                    // a thread's boxes: first bin | last bin << 10 | regular << 31 (the counts and the visit sum below read these)
#pragma unroll
    for (int u = 0; u < PER; ++u) {
        const int i = 64 + tid + u * NTH;
        qq[u] = 0u;
        if (i < n) {
            const float4 bx = sb[i];
            if (sweep_regular(bx)) { const int q0 = bins.of(bx.x); qq[u] = (unsigned)q0 | ((unsigned)bins.of(bx.z) << 10) | 0x80000000u; atomicAdd(&hist[q0], 1); }
        }
    }
    __syncthreads();
    if (wave == 0) {                                                     // exclusive prefix: hist[q] = boxes left of bin q, hist[BINS] = all regular ones
        int mine = 0;
        for (int q = 0; q < YN_SWEEP_BINS / 64; ++q) mine += hist[lane * (YN_SWEEP_BINS / 64) + q];
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        int run = incl - mine;
        for (int q = 0; q < YN_SWEEP_BINS / 64; ++q) { const int k = lane * (YN_SWEEP_BINS / 64) + q; const int cnt = hist[k]; hist[k] = run; run += cnt; }
        if (lane == 63) hist[YN_SWEEP_BINS] = run;
    }
    __syncthreads();
    {
        unsigned long long w = (unsigned long long)(tid < nirr ? n : 0);
#pragma unroll
        for (int u = 0; u < PER; ++u) if (qq[u] >> 31) w += (unsigned long long)(hist[((qq[u] >> 10) & 1023u) + 1] - hist[qq[u] & 1023u]);
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned l2 = (unsigned)__shfl_xor((int)(unsigned)(w & 0xffffffffull), o), h2 = (unsigned)__shfl_xor((int)(unsigned)(w >> 32), o);
            w += ((unsigned long long)h2 << 32) | l2;
        }
        if (lane == 0) atomicAdd(&L.keepm, w);
    }
    __syncthreads();
    const int m = n - 64;
    return L.keepm <= (unsigned long long)m * (unsigned long long)(m - 1) / 8ull;
}

__global__ __launch_bounds__(64 * YN_PRE_W, 8) void nms_prefilter_kernel(const float4* __restrict__ sbox, const int32_t* __restrict__ seg_count,
                                                             const int32_t* __restrict__ seg_off, const int32_t* __restrict__ bucket,
                                                             int N, int C, float thresh, int32_t* __restrict__ keep,
                                                             float4* __restrict__ sbox2, int32_t* __restrict__ bucket2, int32_t* __restrict__ seg_count2,
                                                             const int32_t* __restrict__ seg_order, int32_t* __restrict__ seg_sparse,
                                                             u64* __restrict__ pre_sync, int sliced_ranks,
                                                             const int32_t* __restrict__ large_list, int large_cap, int sweep_slots, int sweep_maxn)
{
    __shared__ PrefilterLds L;
    // grid (B, C + ...): workgroups start in id order, x fastest - every image's LARGEST segment first (seg_order), then the second largest ...:
    // the few 2 000-box segments of a batch, whose workgroups run 10x longer than the rest, no longer start behind 2 000 short ones
    constexpr int NTH = 64 * YN_PRE_W;
    const int b = blockIdx.x;
    int rank = blockIdx.y, z = 0;
    if ((int)blockIdx.y < sliced_ranks * (YN_PRE_Z + 1)) { rank = blockIdx.y / (YN_PRE_Z + 1); z = blockIdx.y % (YN_PRE_Z + 1); }
    else rank = blockIdx.y - sliced_ranks * YN_PRE_Z;
    const int c = seg_order ? seg_order[(size_t)b * C + rank] : rank;
    const int n = seg_count[(size_t)b * C + c];
    const int T = (n + 63) >> 6;
    const bool sliced = rank < sliced_ranks && T - 1 > YN_PRE_W;               // more than one round of band tiles: a round per slice
    const int ZS = sliced ? min(YN_PRE_Z, (T - 1 + YN_PRE_W - 1) / YN_PRE_W) : 1;
    // the sweep's decision: for a sliced segment above 1 024 boxes among the image's first `sweep_slots` listed ones (the sweep's workgroups)
    bool decide = false;
    if (sliced && seg_sparse && sweep_slots > 0 && large_list && nms_colmajor(n)) {
        const int32_t* ll = large_list + (size_t)b * (large_cap + 1);
        const int nl = ll[0] < sweep_slots ? ll[0] : sweep_slots;
        for (int k = 0; k < nl; ++k) decide |= ll[1 + k] == c;
    }
    const bool decider = decide && z == ZS;                                     // (one workgroup behind the band slices)
    if (z >= ZS && !decider) return;
    // (seg_count2 / seg_sparse of a segment: written by ONE workgroup - the one that compacts it.  Two workgroups' plain stores to one word sit in two
    // XCDs' L2s, and which is written back last is anybody's guess: an early `seg_sparse = 0` by slice 0 lost or won against the compacting
    // workgroup's 1 from run to run)
    if (n == 0) { if (threadIdx.x == 0) { seg_count2[(size_t)b * C + c] = 0; if (seg_sparse) seg_sparse[(size_t)b * C + c] = 0; } return; }
    const int off = seg_off[(size_t)b * C + c];
    const float4* sb = sbox + (size_t)b * N + off;
    const int32_t* ids = bucket + (size_t)b * N + off;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)(tid >> 6));
    const size_t slot = (size_t)b * YN_PRE_RANKS + (rank < YN_PRE_RANKS ? rank : 0);
    u64* ticket = pre_sync + slot;
    u64* gs = pre_sync + YN_PRE_TICKETS + slot * pre_sync_words(N);
    if (decider) {
        const bool sp = sweep_spread_out(sb, n, L);
        if (tid == 0) __hip_atomic_store(&gs[0], sp ? 1ull : 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
        float4* cbox = L.cbox[wave];
        float* carea = L.carea[wave];
        // chunk 0 among itself.  Its 64 x 64 tile eight columns per wavefront, the row words put together through LDS (one wavefront on the whole tile:
        // ~4.5 us, the life of the ~2 000 one-chunk workgroups of a batch and the head of every large segment's critical path); then the reference's
        // loop restricted to 64 boxes, wave 0
        float4 bx0 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (lane < n) bx0 = sb[lane];
        {
            u64 part = 0;
            if (wave * 8 < n) part = tile_word_boxes<false>(bx0, bx0, n, 0, 0, thresh, cbox, carea, wave * 8, wave * 8 + 8);
            L.dpart[lane][wave] = (unsigned char)(part >> (8 * wave));
        }
        __syncthreads();
        if (wave == 0) {
            const u64 diag = *reinterpret_cast<const u64*>(L.dpart[lane]);
            const int cnt = min(64, n);
            u64 alive = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
            u64 work = __ballot(((alive >> lane) & 1ull) && (diag & alive));
            const unsigned dlo = (unsigned)(diag & 0xffffffffu), dhi = (unsigned)(diag >> 32);
            while (work) {
                const int i = __ffsll((long long)work) - 1;                                  // alive at its turn: kept
                const unsigned lo_i = (unsigned)__builtin_amdgcn_readlane((int)dlo, i);
                const unsigned hi_i = (unsigned)__builtin_amdgcn_readlane((int)dhi, i);
                alive &= ~(((u64)hi_i << 32) | (u64)lo_i);
                work &= alive & ~(1ull << i);
            }
            if (((alive >> lane) & 1ull) && z == 0) keep[(size_t)b * N + ids[lane]] = 1;
            if (lane == 0) L.keepm = alive;
        }
        if (T == 1) {                                           // nothing beyond chunk 0: no band, no scan, no survivors (most segments of a batch)
            if (tid == 0) { seg_count2[(size_t)b * C + c] = 0; if (seg_sparse) seg_sparse[(size_t)b * C + c] = 0; }
            return;
        }
        __syncthreads();
        const u64 keepm = L.keepm;
        // band (0, ci): which boxes of chunk ci survive chunk 0's kept boxes.  The row boxes are loaded once, the next tile's column
        // boxes are requested before the current tile is evaluated.
        const float4 bx = bx0;
        float4 cb = make_float4(0.f, 0.f, 0.f, 0.f), cbn = cb;
        int ci = 1 + z * YN_PRE_W + wave;
        if (ci < T && ci * 64 + lane < n) cb = sb[ci * 64 + lane];
        for (; ci < T; ci += YN_PRE_W * ZS) {
            const int cn = ci + YN_PRE_W * ZS;
            cbn = make_float4(0.f, 0.f, 0.f, 0.f);
            if (cn < T && cn * 64 + lane < n) cbn = sb[cn * 64 + lane];
            u64 w = keepm ? tile_word_boxes<false>(bx, cb, n, 0, ci, thresh, cbox, carea) : 0ull;
            w = ((keepm >> lane) & 1ull) ? w : 0ull;
            unsigned lo = (unsigned)(w & 0xffffffffu), hi = (unsigned)(w >> 32);
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) { lo |= __shfl_xor(lo, o); hi |= __shfl_xor(hi, o); }
            const int cnt = min(64, n - ci * 64);
            const u64 valid = cnt == 64 ? ~0ull : ((1ull << cnt) - 1ull);
            const u64 sv = valid & ~(((u64)hi << 32) | (u64)lo);
            if (lane == 0) { if (sliced) __hip_atomic_store(&gs[ci], sv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); else L.surv[ci] = sv; }
            cb = cbn;
        }
    }
    bool spread = false;
    if (sliced) {
        // every slice's words out, then its ticket; the last one in reads them all and goes on alone.  The hand-off of DESIGN 4.3d: write-through
        // (agent-scope) stores, retired by the storing wavefront before the barrier, a relaxed ticket, agent-scope loads on the other side - no
        // fence (an agent-scope release here writes the XCD's whole L2 back: 150 us for this launch, measured)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            const u64 t = __hip_atomic_fetch_add(ticket, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            L.last = t == (u64)(ZS + (decide ? 1 : 0) - 1);
            if (L.last) __hip_atomic_store(ticket, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);      // (zero between launches)
        }
        __syncthreads();
        if (!L.last) return;
        for (int k = 1 + tid; k < T; k += NTH) L.surv[k] = __hip_atomic_load(&gs[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (decide) spread = __hip_atomic_load(&gs[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0ull;
    }
    __syncthreads();
    // order-preserving compaction: exclusive prefix of the chunks' survivor counts (thread = a run of Q chunks, then a block scan)
    const int Q = (T - 1 + NTH - 1) / NTH;
    int mine = 0;
    for (int q = 0; q < Q; ++q) { const int ci = 1 + tid * Q + q; if (ci < T) mine += __popcll(L.surv[ci]); }
    L.part[tid] = mine;
    __syncthreads();
    for (int o = 1; o < NTH; o <<= 1) {
        const int v = tid >= o ? L.part[tid - o] : 0;
        __syncthreads();
        L.part[tid] += v;
        __syncthreads();
    }
    int run = L.part[tid] - mine;                           // exclusive
    for (int q = 0; q < Q; ++q) { const int ci = 1 + tid * Q + q; if (ci < T) { L.base[ci] = run; run += __popcll(L.surv[ci]); } }
    const int n2 = L.part[NTH - 1];
    if (tid == NTH - 1) {
        seg_count2[(size_t)b * C + c] = n2;
        // to nms_sweep_kernel: spread out, and a list of survivors that is still large and fits its LDS (else: the dense tiles, as every other segment)
        if (seg_sparse) seg_sparse[(size_t)b * C + c] = (spread && nms_colmajor(n2) && n2 <= sweep_maxn) ? 1 : 0;
    }
    __syncthreads();
    for (int ci = 1 + wave; ci < T; ci += YN_PRE_W) {
        const u64 m = L.surv[ci];
        if ((m >> lane) & 1ull) {
            const int dst = L.base[ci] + __popcll(m & ((1ull << lane) - 1ull));
            sbox2[(size_t)b * N + off + dst] = sb[ci * 64 + lane];
            bucket2[(size_t)b * N + off + dst] = ids[ci * 64 + lane];
        }
    }
}

// ---- sweep: the suppression words of a LARGE segment whose boxes are spread out, without the dense tiles ----------------------------------------
// The bit matrix costs n^2 / 2 pair evaluations whatever the boxes look like.  Two boxes whose x-extents do not intersect cannot suppress one
// another (w = max(1e-28, x-overlap) = 1e-28: inter <= 1e-8 of either area, far below thr * union - shown below for REGULAR boxes), and on the
// benchmark's largest classes almost no pair's do: 2 194 near-point boxes (all kept: 2.4 M pairs, none overlapping), 2 460 full-height strips
// 0.002 wide (3.0 M pairs, ~12 k with intersecting x-extents) - together 1 134 of an image's 1 335 tiles at 416 x 416, 5 500 of 6 000 at 608 x 608.
// A classic broad phase: count-sort the boxes by the bin of their left edge (1 024 bins over the segment's x-range), and for every box visit only
// the boxes whose left edge lies in the bins its own extent covers - every pair with intersecting x-extents is found from the side of its
// left-most box (the bin map is monotone).  Each found pair gets the EXACT predicate of the dense path (`suppressed`, symmetric in its two boxes),
// and a hit sets its bit in the (zeroed) column-major tiles with an atomic OR - resolve reads the same words as after matrix_kernel.
//   REGULAR box: finite coordinates, width > 1e-20, area >= 1e-20.  For a pair with one regular box i and disjoint x-extents: w = 1e-28,
//   h = max(1e-28, dh) with dh <= h_i = a_i / w_i <= 1e20 a_i, so inter <= 1e-8 a_i; un = (a_i + a_j) - inter >= 0.99999999 a_i > 0 (a_j >= 0 or the
//   box is irregular); p = thr * un >= 1e-6 * 0.99 * 1e-20 > 1e-30 and inter < 0.99999 p (thr >= 1e-6, launch condition): suppressed() returns false
//   through its first early-out, exactly as it would in a dense tile.  IRREGULAR boxes (zero / negative / NaN extents: two zero-area boxes give
//   0/0 = NaN -> removed, wherever they are) are tested against every box of the segment.
// WHICH segments: nms_prefilter_kernel estimates the pairs the sweep would visit (sweep_spread_out: bin counts); a segment over a quarter of
// n^2 / 2, with more than 64 irregular boxes or beyond the LDS copy (6 144 boxes) stays with matrix_kernel (seg_sparse = 0).
// Two to eight 1 024-thread workgroups per marked segment (launch_nms_pipeline).
__global__ __launch_bounds__(1024) void nms_sweep_kernel(const float4* __restrict__ sbox, const int32_t* __restrict__ seg_count, const int32_t* __restrict__ seg_off,
                                                         const int32_t* __restrict__ tile_off, int N, int C, float thresh, u64* __restrict__ M, size_t m_stride,
                                                         const int32_t* __restrict__ large_list, int large_cap, const int32_t* __restrict__ seg_sparse, int maxn)
{
    // Behind matrix_kernel, on the segments nms_prefilter_kernel has marked: matrix_kernel has left their tiles out of its work list and zeroed
    // them chip-wide (one workgroup zeroing its own 300 KB took 100-140 k cycles); here: bins (~13 k cycles), exact tests, atomic ORs.
    extern __shared__ __attribute__((aligned(16))) float4 sw_box[];     // [n] the segment's boxes (score order), then [n] ints: box indices grouped by the bin of their left edge
    __shared__ int start[YN_SWEEP_BINS + 2];
    __shared__ int irr[YN_SWEEP_IRR];
    __shared__ int wide[YN_SWEEP_WIDE];
    __shared__ int n_wide;
    __shared__ unsigned xlo_u, xhi_u;
    __shared__ int n_irr;
    // grid (B, workgroups per segment, listed slots) - the IMAGE fastest: workgroups go to the XCDs round robin by their linear id, and with the slot
    // in x (four of them, two with work) every workgroup that had anything to do landed on four of the eight XCDs - 128 at a time on a chip of 256 CUs
    // whatever the split (start / end stamps of all workgroups, round 6)
    const int b = blockIdx.x;
    const int32_t* ll = large_list + (size_t)b * (large_cap + 1);
    if ((int)blockIdx.z >= ll[0]) return;
    const int c = ll[1 + blockIdx.z];
    const int n = seg_count[(size_t)b * C + c];
    if (!nms_colmajor(n) || n > maxn || !seg_sparse[(size_t)b * C + c]) return;
    const float4* sb = sbox + (size_t)b * N + seg_off[(size_t)b * C + c];
    int* sw_order = reinterpret_cast<int*>(sw_box + n);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < YN_SWEEP_BINS + 2; i += 1024) start[i] = 0;
    if (tid == 0) { xlo_u = 0xffffffffu; xhi_u = 0u; n_irr = 0; n_wide = 0; }
    __syncthreads();
    // the boxes into LDS (every later pass and every pair test reads them there); the x-range of the regular ones (ordered-uint min / max), the irregular ones listed
    {
        unsigned mn = 0xffffffffu, mx = 0u;                              // (per thread, then per wavefront: 1 024 atomics on one LDS word serialise - 125 us in the first form)
        for (int i = tid; i < n; i += 1024) {
            const float4 bx = sb[i];
            sw_box[i] = bx;
            if (sweep_regular(bx)) { mn = min(mn, order_bits(bx.x)); mx = max(mx, order_bits(bx.z)); }
            else { const int k = atomicAdd(&n_irr, 1); if (k < YN_SWEEP_IRR) irr[k] = i; }
        }
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) { mn = min(mn, (unsigned)__shfl_xor((int)mn, o)); mx = max(mx, (unsigned)__shfl_xor((int)mx, o)); }
        if (lane == 0) { atomicMin(&xlo_u, mn); atomicMax(&xhi_u, mx); }
    }
    __syncthreads();
    const int nirr = n_irr;                                             // (<= YN_SWEEP_IRR: counted with the same classifier where the segment was marked)
    const SweepBins bins(xlo_u, xhi_u);
    auto bin_of = [&](float x) { return bins.of(x); };
    for (int i = tid; i < n; i += 1024) {
        const float4 bx = sw_box[i];
        if (sweep_regular(bx)) atomicAdd(&start[bin_of(bx.x) + 2], 1);  // counts, two slots up
    }
    __syncthreads();
    if (wave == 0) {                                                     // exclusive prefix over the bins (16 per lane + a wave scan)
        int mine = 0;
        for (int q = 0; q < YN_SWEEP_BINS / 64; ++q) mine += start[2 + lane * (YN_SWEEP_BINS / 64) + q];
        int incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int v = __shfl_up(incl, o); if (lane >= o) incl += v; }
        int run = incl - mine;
        for (int q = 0; q < YN_SWEEP_BINS / 64; ++q) { const int k = 2 + lane * (YN_SWEEP_BINS / 64) + q; const int cnt = start[k]; start[k] = run; run += cnt; }
    }
    __syncthreads();
    // start[q + 2] = first index of bin q; the scatter bumps it to the bin's end: afterwards first(q) = start[q + 1] (start[1] = 0), end(q) = start[q + 2]
    for (int i = tid; i < n; i += 1024) {
        const float4 bx = sw_box[i];
        if (sweep_regular(bx)) sw_order[atomicAdd(&start[bin_of(bx.x) + 2], 1)] = i;
    }
    __syncthreads();
    u64* Ms = M + (size_t)b * m_stride + (size_t)tile_off[(size_t)b * (C + 1) + c] * 64;      // (zeroed by matrix_kernel)
    // the pair test: matrix_kernel's division-free decision first (tile_word_boxes: inter against thr * union with a 1e-5 guard band - ~25 instructions;
    // `suppressed` inlined whole is ~100, and this loop is VALU-bound: 16 wavefronts on four SIMDs), the exact predicate only inside the band / for
    // union <= 0 / NaN.  A sure decision is implied by the exact arithmetic (rounding is monotonic), so the bits are those of the dense tiles.
    const float c_hi = thresh * 1.00001f, c_lo = thresh * 0.99999f;
    auto test = [&](int i, int j, const float4 bi, float ai, const float4 bj) {
        const float aj = (bj.z - bj.x) * (bj.w - bj.y);
        const float w = vmax_raw(1e-28f, vmin_raw(bi.z, bj.z) - vmax_raw(bi.x, bj.x));
        const float h = vmax_raw(1e-28f, vmin_raw(bi.w, bj.w) - vmax_raw(bi.y, bj.y));
        const float inter = w * h;
        const float un = (ai + aj) - inter;
        const float hi = un * c_hi, lo = un * c_lo;
        bool hit;
        if (lo > 1e-30f && inter < lo) return;                           // surely kept (almost every visit)
        if (lo > 1e-30f && inter > hi) hit = true;
        else hit = suppressed(bi, ai, bj, aj, thresh, 0);
        if (hit) {
            const int r = i < j ? i : j, t = i < j ? j : i;
            atomicOr(Ms + ctile_off(r >> 6, t >> 6) + (r & 63), 1ull << (t & 63));
        }
    };
    // SIXTEEN lanes per box, over the boxes whose left edge lies in the bins its extent covers: from its OWN bin only the later ones (the earlier ones
    // find this box themselves), from later bins all (their own sweep starts at their bin: it never looks back).  A visit is ~100 instructions
    // (index, box, the exact test, the tile address of a hit), and a wavefront runs as long as its longest lane: a thread per box (first forms) left
    // three lanes of four idle - 131 k cycles for 50 k visits.  The segment's boxes are dealt to the segment's workgroups (blockIdx.y; each has
    // binned the whole segment itself).
    const int kz = blockIdx.y, KZ = gridDim.y;
    const int grp = tid >> 4, gl = tid & 15;
    for (int i = kz + KZ * grp; i < n; i += 64 * KZ) {                    // (box i belongs to workgroup i % KZ)
        const float4 bi = sw_box[i];
        if (!sweep_regular(bi)) continue;
        const float ai = (bi.z - bi.x) * (bi.w - bi.y);
        const int q0 = bin_of(bi.x), q1 = bin_of(bi.z);
        const int p0 = start[q0 + 1], pm = start[q0 + 2], p1 = start[q1 + 2];
        if (p1 - p0 > YN_SWEEP_WIDE_VISITS) {                            // a WIDE box: listed for the whole workgroup below (sixteen lanes on 5 000 visits: the kernel's long pole,
            int k = 0;                                                   // whatever the split - 136 us at 608 x 608)
            if (gl == 0) k = atomicAdd(&n_wide, 1);
            k = __shfl(k, 0, 16);
            if (k < YN_SWEEP_WIDE) { if (gl == 0) wide[k] = i; continue; }
        }
        for (int p = p0 + gl; p < p1; p += 16) {
            const int j = sw_order[p];
            if (p < pm ? j > i : true) test(i, j, bi, ai, sw_box[j]);
        }
    }
    __syncthreads();
    {
        const int nw = n_wide < YN_SWEEP_WIDE ? n_wide : YN_SWEEP_WIDE;
        for (int k = 0; k < nw; ++k) {
            const int i = wide[k];
            const float4 bi = sw_box[i];
            const float ai = (bi.z - bi.x) * (bi.w - bi.y);
            const int q0 = bin_of(bi.x), q1 = bin_of(bi.z);
            const int p0 = start[q0 + 1], pm = start[q0 + 2], p1 = start[q1 + 2];
            for (int p = p0 + tid; p < p1; p += 1024) {
                const int j = sw_order[p];
                if (p < pm ? j > i : true) test(i, j, bi, ai, sw_box[j]);
            }
        }
    }
    for (int k = 0; k < nirr; ++k) {                                     // irregular boxes: against everything (each pair once: regular partners, and later irregular ones)
        const int i = irr[k];
        const float4 bi = sw_box[i];
        const float ai = (bi.z - bi.x) * (bi.w - bi.y);
        for (int j = kz + KZ * tid; j < n; j += 1024 * KZ) {
            if (j == i) continue;
            const float4 bj = sw_box[j];
            if (sweep_regular(bj) || j > i) test(i, j, bi, ai, bj);
        }
    }
}

// tile offsets of the prefiltered segments (one thread per image: C is small): tile_off2 = where a segment's tiles are STORED (every segment),
// work_off = the same prefix over the segments matrix_kernel evaluates (all but the ones nms_sweep_kernel has marked)
__global__ __launch_bounds__(64) void nms_tile_off_kernel(const int32_t* __restrict__ seg_count2, int C, int32_t* __restrict__ tile_off2, const int32_t* __restrict__ seg_sparse,
                                                          int32_t* __restrict__ work_off)
{
    // one wavefront per image, a class per lane, a wave scan (one thread walking the 80 classes: 6.5 us of dependent loads)
    const int b = blockIdx.x, lane = threadIdx.x;
    int tiles = 0, wtiles = 0;
    for (int c0 = 0; c0 < C; c0 += 64) {
        const int c = c0 + lane;
        int t = 0, w = 0;
        if (c < C) {
            const int T = (seg_count2[(size_t)b * C + c] + 63) >> 6;
            t = T * (T + 1) / 2;
            w = (seg_sparse && seg_sparse[(size_t)b * C + c]) ? 0 : t;
        }
        int ti = t, wi = w;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const int a = __shfl_up(ti, o), d = __shfl_up(wi, o); if (lane >= o) { ti += a; wi += d; } }
        if (c < C) {
            tile_off2[(size_t)b * (C + 1) + c] = tiles + ti - t;
            if (work_off) work_off[(size_t)b * (C + 1) + c] = wtiles + wi - w;
        }
        tiles += __shfl(ti, 63);
        wtiles += __shfl(wi, 63);
    }
    if (lane == 0) {
        tile_off2[(size_t)b * (C + 1) + C] = tiles;
        if (work_off) work_off[(size_t)b * (C + 1) + C] = wtiles;
    }
}

// ---- single-class entry (YOLONano.nms): one segment = items 0..n-1 ------------------------------------
__global__ __launch_bounds__(1024) void single_sort_kernel(const float* __restrict__ dets, const float* __restrict__ scores, int n,
                                                           int32_t* __restrict__ ids, float4* __restrict__ sbox, u64* __restrict__ gscratch)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char sort_lds[];
    if (gscratch) sort_segment<false>(dets, scores, ids, false, n, nms_pow2(n), gscratch, sbox);
    else sort_segment<true>(dets, scores, ids, false, n, nms_pow2(n), reinterpret_cast<u64*>(sort_lds), sbox);
}

template <bool DIOU>
__global__ __launch_bounds__(64) void single_matrix_kernel(const float4* __restrict__ sbox, int n, float thresh, u64* __restrict__ M)
{
    __shared__ float4 cbox[64];
    __shared__ float carea[64];
    const int T = (n + 63) >> 6;
    const int total = T * (T + 1) / 2;
    for (int t = blockIdx.x; t < total; t += gridDim.x) {
        int rem = t, ri = 0;
        while (rem >= T - ri) { rem -= T - ri; ++ri; }
        matrix_tile<DIOU>(sbox, n, T, ri, ri + rem, thresh, M, cbox, carea);
    }
}

__global__ __launch_bounds__(256) void single_resolve_kernel(const int32_t* __restrict__ ids, int n, const u64* __restrict__ M,
                                                              int32_t* __restrict__ pick_list, int32_t* __restrict__ count)
{
    __shared__ ResolveLds L;
    const int picked = n > 0 ? resolve_segment<2>(ids, n, M, nullptr, pick_list, L) : 0;
    if (threadIdx.x == 0) *count = picked;
}

// kept candidates of image b, ascending candidate index (np.where(keep > 0), models/yolo_nano.py:274-277)
__global__ __launch_bounds__(1024) void compact_kernel(const float* __restrict__ boxes, const float* __restrict__ scores,
                                                        const int32_t* __restrict__ cls, const int32_t* __restrict__ keep, int N,
                                                        float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                        int32_t* __restrict__ out_cls, int32_t* __restrict__ out_index,
                                                        int32_t* __restrict__ count, const unsigned* __restrict__ ovf, unsigned* __restrict__ ovf_host)
{
    // 32 chunks of 1024 candidates per pass: every keep flag of the pass is requested in one batch, the per-chunk wavefront counts go to
    // LDS, ONE barrier, then every thread derives the positions of its (up to 32) kept candidates.  (First version: load, ballot,
    // barrier x 3 per chunk — a 1.2 us serial step 11 / 23 times per image at 416 / 608: 15 / 27 us of the bs = 1 chain.)
    constexpr int IT = 32;
    __shared__ int wave_sums[IT][16];
    __shared__ int part[8];
    __shared__ int base, total_next;
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) base = 0;
    __syncthreads();
    for (int n0 = 0; n0 < N; n0 += IT * 1024) {
        int kf[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int n = n0 + it * 1024 + threadIdx.x;
            kf[it] = keep[(size_t)b * N + (n < N ? n : N - 1)];
        }
        unsigned flags = 0;
        int before[IT];
#pragma unroll
        for (int it = 0; it < IT; ++it) {
            const int n = n0 + it * 1024 + threadIdx.x;
            const int f = (n < N && kf[it]) ? 1 : 0;
            const unsigned long long bal = __ballot(f);
            before[it] = __popcll(bal & ((1ull << lane) - 1ull));
            if (lane == 0) wave_sums[it][wave] = __popcll(bal);
            flags |= (unsigned)f << it;
        }
        __syncthreads();
        // exclusive scan of the 512 (chunk, wavefront) counts in candidate order: wavefront scan (shuffles), then the 8 wavefront totals
        int v = 0, incl = 0;
        if (threadIdx.x < IT * 16) {
            v = (&wave_sums[0][0])[threadIdx.x];
            incl = v;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const int t2 = __shfl_up(incl, o); if (lane >= o) incl += t2; }
            if (lane == 63) part[wave] = incl;
        }
        __syncthreads();
        if (threadIdx.x < IT * 16) {
            int add = 0;
#pragma unroll
            for (int w = 0; w < 8; ++w) add += (w < wave) ? part[w] : 0;
            (&wave_sums[0][0])[threadIdx.x] = base + add + incl - v;          // position of the first kept candidate of (chunk, wavefront)
            if (threadIdx.x == IT * 16 - 1) total_next = base + add + incl;
        }
        __syncthreads();
        // the copies in groups of eight chunks: all loads of a group are issued before its first store (a load next to a store of
        // unknown aliasing waits for it: up to 32 dependent round trips per thread otherwise)
#pragma unroll
        for (int g8 = 0; g8 < IT; g8 += 8) {
            float4 bx[8]; float sc[8]; int cl[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int it = g8 + k;
                const int n = n0 + it * 1024 + threadIdx.x;
                const size_t src = (size_t)b * N + (((flags >> it) & 1u) ? n : 0);
                bx[k] = *reinterpret_cast<const float4*>(boxes + src * 4);
                sc[k] = scores[src];
                cl[k] = cls[src];
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int it = g8 + k;
                if ((flags >> it) & 1u) {
                    const int n = n0 + it * 1024 + threadIdx.x;
                    const size_t dst = (size_t)b * N + wave_sums[it][wave] + before[it];
                    *reinterpret_cast<float4*>(out_boxes + dst * 4) = bx[k];
                    out_scores[dst] = sc[k];
                    out_cls[dst] = cl[k];
                    if (out_index) out_index[dst] = n;
                }
            }
        }
        __syncthreads();                                    // every thread has read this pass's positions
        if (threadIdx.x == 0) base = total_next;
        __syncthreads();
    }
    // range guard of the split-f16 family, delivered with the result every caller reads anyway: a NEGATIVE count (-1 - kept) says an
    // activation left the split's range somewhere in the network that produced these candidates (yn_range_status; re-run under yn_exact_f32)
    // The same fact out of band (advisor, round 4): one word of pinned host memory per handle, which every later C-ABI call of the handle
    // checks without synchronising - it returns YN_STATUS_RANGE until yn_range_status() acknowledges (and clears) the flag.
    if (threadIdx.x == 0) {
        const bool flagged = ovf && *ovf;
        count[b] = flagged ? -1 - base : base;
        if (flagged && ovf_host) __hip_atomic_store(ovf_host, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

int nms_max_segment() { return 64 * YN_RESOLVE_MAX_T; }

// The `.to('cpu').numpy()` hand-over of models/yolo_nano.py:370-376 for a whole batch: the kept rows of all B images as ONE
// contiguous record list rec[total][6] = x1, y1, x2, y2, score, class (image order, ascending candidate order inside an
// image) + offsets[B+1], so the host needs two copies per batch.  Block = one image; its base is the sum of the earlier counts.
__global__ __launch_bounds__(256) void pack_kernel(const float* __restrict__ boxes, const float* __restrict__ scores, const int32_t* __restrict__ cls,
                                                   const int32_t* __restrict__ count, int B, int N, float* __restrict__ rec, int32_t* __restrict__ offsets)
{
    __shared__ int base_s, total_s;
    const int b = blockIdx.x;
    if (threadIdx.x < 64) {
        int mine = 0, all = 0;
        int bad = 0;
        for (int i = threadIdx.x; i < B; i += 64) { int c = count[i]; if (c < 0) { bad = 1; c = -1 - c; } all += c; if (i < b) mine += c; }
        for (int o = 32; o > 0; o >>= 1) { mine += __shfl_xor(mine, o); all += __shfl_xor(all, o); bad |= __shfl_xor(bad, o); }
        if (threadIdx.x == 0) { base_s = mine; total_s = bad ? -1 - all : all; }      // a flagged batch (compact_kernel) keeps its mark: offsets[B] < 0
    }
    __syncthreads();
    const int base = base_s, k = count[b] < 0 ? -1 - count[b] : count[b];
    if (threadIdx.x == 0) { offsets[b] = base; if (b == 0) offsets[B] = total_s; }
    for (int i = threadIdx.x; i < k; i += 256) {
        const size_t src = (size_t)b * N + i;
        const float4 bx = *reinterpret_cast<const float4*>(boxes + src * 4);
        float* r = rec + (size_t)(base + i) * 6;            // 24-byte records: 8-byte aligned
        *reinterpret_cast<float2*>(r) = make_float2(bx.x, bx.y);
        *reinterpret_cast<float2*>(r + 2) = make_float2(bx.z, bx.w);
        *reinterpret_cast<float2*>(r + 4) = make_float2(scores[src], (float)cls[src]);
    }
}

void launch_pack(const float* boxes, const float* scores, const int32_t* cls, const int32_t* count, int B, int N, float* rec, int32_t* offsets, hipStream_t s)
{
    hipLaunchKernelGGL(pack_kernel, dim3(B), dim3(256), 0, s, boxes, scores, cls, count, B, N, rec, offsets);
}

size_t nms_pre_sync_words(int B, int N) { return YN_PRE_TICKETS + (size_t)B * YN_PRE_RANKS * pre_sync_words(N); }
size_t nms_matrix_words_per_image(int N, int C)
{
    const size_t Tsum = (size_t)(N + 63) / 64 + C;          // sum_c ceil(n_c/64) <= N/64 + C
    return 64 * (Tsum * (Tsum + 1) / 2);
}

static void set_sort_attr()
{
    static unsigned long long done = 0;
    if (!attr_pending(done)) return;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, YN_SORT_LARGE * 8);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(single_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, YN_SORT_LARGE * 8);
}

void launch_nms_pipeline(const float* boxes, const float* scores, const int32_t* cls, int B, int N, int C,
                         float nms_thresh, int diou, const NmsWork& wk,
                         float* out_boxes, float* out_scores, int32_t* out_cls, int32_t* out_index, int32_t* count,
                         hipStream_t s, const NmsHook* hook)
{
    set_sort_attr();
    const int large_cap = wk.large_cap;
    auto mark = [&](const char* k) { if (hook && hook->fn) hook->fn(hook->ctx, k); };
    // YN_DBG_NMS_SKIP (timing ablation only, outputs are wrong): bit0 sort, bit1 matrix, bit2 resolve
    static const int skip = getenv("YN_DBG_NMS_SKIP") ? atoi(getenv("YN_DBG_NMS_SKIP")) : 0;
    float4* sbox = reinterpret_cast<float4*>(wk.sbox);
    u64* M = reinterpret_cast<u64*>(wk.matrix);
    static const int fuse_bs = getenv("YN_NMS_FUSE_BUCKET") ? atoi(getenv("YN_NMS_FUSE_BUCKET")) : 1;       // A/B: 0 = bucket_kernel + sort_kernel also for few segments
    // "few segments" (one to three images at 80 classes): everything of an image's NMS on big workgroups that are all resident at once.  Not for
    // maps with more than 16 K candidates per image (608 x 608: one class of the benchmark's images holds ~10 000 boxes): there the 16 384-key
    // bitonic network of the one-workgroup sort is 60 us of a 0.57 ms call, and the chunked sort + merge (three launches) is shorter
    static const int few_n = getenv("YN_NMS_FEW_N") ? atoi(getenv("YN_NMS_FEW_N")) : 16384;
    const bool few = (long)B * C <= 256 && N <= few_n;
    const int32_t* seg_order = wk.seg_order;                // bucket_kernel's size ranking; the fused kernel does not produce one (few segments: nothing to order)
    if (few && fuse_bs && wk.ctr && !(skip & 1)) {
        seg_order = nullptr;
        static unsigned long long attr_bs = 0;
        if (attr_pending(attr_bs)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(bucket_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, YN_SORT_LARGE * 8);
        mark("bucket_sort_kernel");
        hipLaunchKernelGGL(bucket_sort_kernel, dim3(C, B), dim3(1024), YN_SORT_LARGE * 8, s, boxes, scores, cls, N, C, wk.seg_count, wk.seg_off, wk.tile_off, wk.bucket,
                           wk.keep, sbox, M, wk.matrix_stride, wk.ctr);
    } else {
    mark("bucket_kernel");
    hipLaunchKernelGGL(bucket_kernel, dim3(B), dim3(1024), 2 * C * sizeof(int32_t), s, cls, N, C, wk.seg_count, wk.seg_off, wk.tile_off, wk.bucket, wk.keep,
                       wk.large_list, large_cap, YN_SORT_SMALL, wk.seg_order);
    static const int chunked_env = getenv("YN_NMS_SORT_CHUNKS") ? atoi(getenv("YN_NMS_SORT_CHUNKS")) : 1;   // A/B: 0 = one 1024-thread workgroup per large segment
    const bool chunks = !few && chunked_env && wk.large_list && (size_t)N <= wk.matrix_stride;
    mark(chunks ? "sort_chunk_kernel" : "sort_kernel");
    if (!(skip & 1)) {
    if (few) {
        // few segments (bs <= 3 at 80 classes): every one gets a 1024-thread / 128 KB-LDS workgroup, all resident at once - one launch
        // whose duration is the largest segment's sort instead of the small launch followed by the large one (64 -> 35 us at bs = 1)
        hipLaunchKernelGGL(sort_kernel, dim3(B, C), dim3(1024), YN_SORT_LARGE * 8, s, boxes, scores, wk.seg_count, wk.seg_off, wk.bucket, sbox,
                           N, C, 0, YN_SORT_LARGE, (u64*)nullptr, (size_t)0, (const int32_t*)nullptr, 0, (const int32_t*)wk.seg_order);
    } else {
    if (chunks) {
        // small segments whole + the large segments' 1024-box chunks in one launch of 256-thread workgroups, then the merge (keys in the not yet used matrix area)
        const int slots = N > YN_SORT_SMALL ? nms_chunk_slots(N, large_cap) : 0;
        hipLaunchKernelGGL(sort_chunk_kernel, dim3(B, C + slots), dim3(256), YN_SORT_SMALL * 8, s, boxes, scores, wk.seg_count, wk.seg_off, wk.bucket, sbox,
                           N, C, (const int32_t*)wk.seg_order, (const int32_t*)wk.large_list, large_cap, M, wk.matrix_stride);
        if (N > YN_SORT_SMALL) {
            mark("sort_merge_kernel");
            hipLaunchKernelGGL(sort_merge_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, boxes, wk.seg_count, wk.seg_off, wk.bucket, sbox,
                               N, C, (const int32_t*)wk.large_list, large_cap, (const u64*)M, wk.matrix_stride);
        }
    } else {
    hipLaunchKernelGGL(sort_kernel, dim3(B, C), dim3(256), YN_SORT_SMALL * 8, s, boxes, scores, wk.seg_count, wk.seg_off, wk.bucket, sbox,
                       N, C, 0, YN_SORT_SMALL, (u64*)nullptr, (size_t)0, (const int32_t*)nullptr, 0, (const int32_t*)wk.seg_order);
    if (N > YN_SORT_SMALL)                                  // only the (few) listed large segments get a 128 KB-LDS workgroup
        hipLaunchKernelGGL(sort_kernel, dim3(large_cap, B), dim3(1024), YN_SORT_LARGE * 8, s, boxes, scores, wk.seg_count, wk.seg_off, wk.bucket, sbox,
                           N, C, YN_SORT_SMALL, YN_SORT_LARGE, (u64*)nullptr, (size_t)0, (const int32_t*)wk.large_list, large_cap, (const int32_t*)nullptr);
    }
    }
    if (N > YN_SORT_LARGE && !chunks)                       // at most one such segment per image: keys in the (not yet used) matrix area
        hipLaunchKernelGGL(sort_kernel, dim3(large_cap, B), dim3(1024), 0, s, boxes, scores, wk.seg_count, wk.seg_off, wk.bucket, sbox,
                           N, C, YN_SORT_LARGE, 1 << 30, M, wk.matrix_stride, (const int32_t*)wk.large_list, large_cap, (const int32_t*)nullptr);
    }
    }
    const int32_t* m_count = wk.seg_count;
    const int32_t* m_toff = wk.tile_off;
    const int32_t* m_ids = wk.bucket;
    const float4* m_box = sbox;
    const int prefilter_env = wk.prefilter;
    bool sweep = false;
    int sweep_maxn = 0, sweep_slots = 0;
    // (small batches: the prefilter is two more launches in a serial chain - bs = 1 latency 0.69 -> 0.72 ms - for chip time nobody else wants)
    if (!diou && wk.sbox2 && (prefilter_env == 2 || (prefilter_env == 1 && B >= 4))) {
        mark("nms_prefilter_kernel");
        // large segments whose boxes are spread out: their suppression words from a sweep over bins of the left edges instead of the dense tiles
        // (nms_sweep_kernel): decided at the end of the prefilter, zeroed by matrix_kernel, filled in behind it.
        // The sweep's workgroups carry the segment in LDS (20 bytes per box): 60 KB - two per CU - for maps of up to 16 K candidates (a class above 3 072 boxes stays
        // dense there), 120 KB beyond (608 x 608: ~5 000-box classes); the first four listed segments of an image only (the list is by size: a fifth class above
        // 1 024 boxes is rare, and every listed slot is a workgroup with that LDS to schedule whether it has work or not)
        sweep = wk.sweep && !few && wk.seg_sparse && wk.work_off && wk.large_list && large_cap > 0 && N > YN_SORT_SMALL && nms_thresh >= 1e-6f && !(skip & 2);
        sweep_maxn = N <= 16384 ? YN_SWEEP_MAXN / 2 : YN_SWEEP_MAXN;
        sweep_slots = large_cap < 4 ? large_cap : 4;
        const int pre_ranks = (seg_order && wk.pre_sync && (size_t)B * YN_PRE_RANKS <= YN_PRE_TICKETS) ? (C < YN_PRE_RANKS ? C : YN_PRE_RANKS) : 0;      // large segments: band sliced over YN_PRE_Z workgroups
        hipLaunchKernelGGL(nms_prefilter_kernel, dim3(B, C + pre_ranks * YN_PRE_Z), dim3(64 * YN_PRE_W), 0, s, sbox, wk.seg_count, wk.seg_off, wk.bucket, N, C, nms_thresh, wk.keep,
                           reinterpret_cast<float4*>(wk.sbox2), wk.bucket2, wk.seg_count2, seg_order, wk.seg_sparse, reinterpret_cast<u64*>(wk.pre_sync), pre_ranks,
                           (const int32_t*)wk.large_list, large_cap, sweep ? sweep_slots : 0, sweep_maxn);
        m_count = wk.seg_count2; m_toff = wk.tile_off2; m_ids = wk.bucket2; m_box = reinterpret_cast<const float4*>(wk.sbox2);
        mark("nms_tile_off_kernel");
        hipLaunchKernelGGL(nms_tile_off_kernel, dim3(B), dim3(64), 0, s, wk.seg_count2, C, wk.tile_off2, (const int32_t*)(sweep ? wk.seg_sparse : nullptr), sweep ? wk.work_off : nullptr);
    }
    int G = 4096 / (B > 0 ? B : 1);                         // x4 wavefronts per block
    if (G < 32) G = 32;
    if (G > 2048) G = 2048;
    mark(diou ? "matrix_kernel<true>" : "matrix_kernel<false>");
    if (skip & 2) {}
    else if (diou) hipLaunchKernelGGL(matrix_kernel<true>, dim3(G, B), dim3(256), 0, s, m_box, m_count, wk.seg_off, m_toff, N, C, nms_thresh, M, wk.matrix_stride,
                                      (const int32_t*)nullptr, (const int32_t*)nullptr, (const int32_t*)nullptr, 0);
    else      hipLaunchKernelGGL(matrix_kernel<false>, dim3(G, B), dim3(256), 0, s, m_box, m_count, wk.seg_off, m_toff, N, C, nms_thresh, M, wk.matrix_stride,
                                 (const int32_t*)(sweep ? wk.work_off : nullptr), (const int32_t*)(sweep ? wk.seg_sparse : nullptr), (const int32_t*)wk.large_list, large_cap);
    if (sweep) {
        mark("nms_sweep_kernel");
        // workgroups per segment: about as many working ones (two marked segments per image, typically) as the chip holds - two per CU with the 60 KB
        // form, one with the 120 KB form.  Measured (us; 416 bs 32 / 608 bs 32 / 0.5x bs 128 / 416 bs 8): 2 -> 32.8 / 122 / 29.1 / -, 4 -> 29.3 / 88.8 / 34.0 / 18.6,
        // 8 -> 28.0 / 97.2 / 46.2 / 15.7 (every workgroup repeats the binning, ~13 k cycles)
        int sweep_split = (sweep_maxn * 20 <= 64 * 1024 ? 256 : 128) / (B > 0 ? B : 1);
        sweep_split = sweep_split < 2 ? 2 : (sweep_split > 8 ? 8 : sweep_split);
        static unsigned long long attr_sw = 0;
        if (attr_pending(attr_sw)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sweep_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, YN_SWEEP_MAXN * 20);
        hipLaunchKernelGGL(nms_sweep_kernel, dim3(B, sweep_split, sweep_slots), dim3(1024), (size_t)sweep_maxn * 20, s, m_box, m_count, wk.seg_off, m_toff, N, C, nms_thresh,
                           M, wk.matrix_stride, (const int32_t*)wk.large_list, large_cap, (const int32_t*)wk.seg_sparse, sweep_maxn);
    }
    mark("resolve_kernel");
    const bool split = N > YN_SORT_SMALL && wk.large_list && large_cap > 0;
    if (!(skip & 4) && (long)B * C <= 256) {
        // few segments (bs <= 3 at 80 classes): all of them on the 512-thread kernel in ONE launch — the short walks ride along with the long
        // ones instead of preceding them (one image: -20 us)
        hipLaunchKernelGGL(resolve_large_kernel, dim3(C, B), dim3(512), 0, s, m_count, wk.seg_off, m_toff, m_ids, N, C, M, wk.matrix_stride,
                           wk.keep, (const int32_t*)nullptr, 0, 0);
    } else if (!(skip & 4)) {
        hipLaunchKernelGGL(resolve_kernel, dim3(B, C), dim3(256), 0, s, m_count, wk.seg_off, m_toff, m_ids, N, C, M, wk.matrix_stride, wk.keep,
                           split ? YN_SORT_SMALL : 1 << 30, seg_order);
        if (split) hipLaunchKernelGGL(resolve_large_kernel, dim3(B, large_cap), dim3(512), 0, s, m_count, wk.seg_off, m_toff, m_ids, N, C, M, wk.matrix_stride,
                                      wk.keep, (const int32_t*)wk.large_list, large_cap, YN_SORT_SMALL);
    }
    mark("compact_kernel");
    hipLaunchKernelGGL(compact_kernel, dim3(B), dim3(1024), 0, s, boxes, scores, cls, wk.keep, N, out_boxes, out_scores, out_cls, out_index, count, wk.ovf, wk.ovf_host);
}

// scratch: ids[n] int32, sbox[n] float4, M[nms_matrix_words_per_image(n,1)] u64 — all provided by the handle
void launch_nms_single(const float* dets, const float* scores, int n, float thresh, int diou,
                       int32_t* ids_scratch, float* sbox_scratch, void* matrix_scratch, int32_t* keep, int32_t* count, hipStream_t s)
{
    set_sort_attr();
    float4* sbox = reinterpret_cast<float4*>(sbox_scratch);
    u64* M = reinterpret_cast<u64*>(matrix_scratch);
    if (n > 0) {
        const int P = nms_pow2(n);
        const int thr = P >= 2048 ? 1024 : 256;
        if (P <= YN_SORT_LARGE) hipLaunchKernelGGL(single_sort_kernel, dim3(1), dim3(thr), (size_t)P * 8, s, dets, scores, n, ids_scratch, sbox, (u64*)nullptr);
        else hipLaunchKernelGGL(single_sort_kernel, dim3(1), dim3(1024), 0, s, dets, scores, n, ids_scratch, sbox, M);
        const int T = (n + 63) / 64;
        int G = T * (T + 1) / 2;
        if (G > 8192) G = 8192;
        if (diou) hipLaunchKernelGGL(single_matrix_kernel<true>, dim3(G), dim3(64), 0, s, sbox, n, thresh, M);
        else      hipLaunchKernelGGL(single_matrix_kernel<false>, dim3(G), dim3(64), 0, s, sbox, n, thresh, M);
    }
    hipLaunchKernelGGL(single_resolve_kernel, dim3(1), dim3(256), 0, s, ids_scratch, n, M, keep, count);
}

// -------------------------------------------------------------------------------------------------
// ValTransforms on the device (data/transforms.py:59-70, 73-119, 394-398, 445-458): uint8 HWC BGR image -> letterbox resize
// (cv2.resize INTER_LINEAR, 8-bit fixed point: modules/imgproc/src/resize.cpp, restated in oracle/preprocess.py — the oracle's
// header says why this path's parity is UNPINNED) -> mean padding -> /255, -mean, /std -> RGB CHW float32, written straight
// into one image slot of the network input.  Thread = one output pixel (three channels); HBM-bound: 3 bytes read per resized
// pixel (4 taps from L1/L2), 12 bytes written.  mode: 0 copy, 1 exact 2:1 reduction (2x2 box), 2 linear.
// -------------------------------------------------------------------------------------------------
struct PrepArgs {
    const unsigned char* img; int h0, w0;       // source
    int rw, rh, left, top, side, mode;          // resized extent, placement inside the side x side square
    float mean[3], std[3];                      // BGR order, as the reference passes them
    float* out;                                 // [3][side][side], channel 0 = R
};

__device__ __forceinline__ void preprocess_pixel(const PrepArgs& a, int i)
{
    if (i >= a.side * a.side) return;
    const int y = i / a.side, x = i - y * a.side;
    const int ry = y - a.top, rx = x - a.left;
    float v[3];
    if (ry >= 0 && ry < a.rh && rx >= 0 && rx < a.rw) {
        int u[3];
        if (a.mode == 0) {
            const unsigned char* p = a.img + ((size_t)ry * a.w0 + rx) * 3;
            u[0] = p[0]; u[1] = p[1]; u[2] = p[2];
        } else if (a.mode == 1) {
            const unsigned char* p = a.img + ((size_t)(2 * ry) * a.w0 + 2 * rx) * 3;
            const unsigned char* q = p + (size_t)a.w0 * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) u[c] = (p[c] + p[3 + c] + q[c] + q[3 + c] + 2) >> 2;
        } else {
            // resizeGeneric_ (ksize 2): fx = (float)((dx + 0.5) * scale - 0.5), scale = 1 / (dsize / ssize) in double
            const double scx = 1.0 / ((double)a.rw / (double)a.w0), scy = 1.0 / ((double)a.rh / (double)a.h0);
            float fx = (float)(((double)rx + 0.5) * scx - 0.5), fy = (float)(((double)ry + 0.5) * scy - 0.5);
            int sx = (int)floorf(fx), sy = (int)floorf(fy);
            fx -= (float)sx; fy -= (float)sy;
            if (sx < 0) { fx = 0.0f; sx = 0; }
            if (sx >= a.w0 - 1) { fx = 0.0f; sx = a.w0 - 1; }
            const int a0 = __float2int_rn((1.0f - fx) * 2048.0f), a1 = __float2int_rn(fx * 2048.0f);   // cvRound -> short
            const int b0 = __float2int_rn((1.0f - fy) * 2048.0f), b1 = __float2int_rn(fy * 2048.0f);
            const int sx1 = min(sx + 1, a.w0 - 1);
            const int r0 = min(max(sy, 0), a.h0 - 1), r1 = min(max(sy + 1, 0), a.h0 - 1);
            const unsigned char* p0 = a.img + (size_t)r0 * a.w0 * 3;
            const unsigned char* p1 = a.img + (size_t)r1 * a.w0 * 3;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const int s0 = p0[sx * 3 + c] * a0 + p0[sx1 * 3 + c] * a1;          // HResizeLinear (scale 2048)
                const int s1 = p1[sx * 3 + c] * a0 + p1[sx1 * 3 + c] * a1;
                const int r = (((b0 * (s0 >> 4)) >> 16) + ((b1 * (s1 >> 4)) >> 16) + 2) >> 2;   // VResizeLinear 8u
                u[c] = min(max(r, 0), 255);
            }
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = (float)u[c];
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) v[c] = a.mean[c] * 255.0f;                     // Resize.mean = [v * 255 for v in mean]
    }
    const size_t plane = (size_t)a.side * a.side;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float t = v[c] / 255.0f;                                                   // Normalize: image /= 255.; -= mean; /= std
        t = t - a.mean[c];
        t = t / a.std[c];
        a.out[(size_t)(2 - c) * plane + i] = t;                                    // ToTensor: BGR -> RGB, HWC -> CHW
    }
}

__global__ __launch_bounds__(256) void preprocess_kernel(PrepArgs a) { preprocess_pixel(a, blockIdx.x * 256 + threadIdx.x); }

// up to PREP_MAX images per launch (descriptors by value in the kernel arguments): blockIdx.y = image
constexpr int PREP_MAX = 32;
struct PrepImg { const unsigned char* img; int h0, w0, rw, rh, left, top; };
struct PrepBatchArgs { PrepImg im[PREP_MAX]; int side; float mean[3], std[3]; float* out; };
__global__ __launch_bounds__(256) void preprocess_batch_kernel(PrepBatchArgs b)
{
    const PrepImg& d = b.im[blockIdx.y];
    PrepArgs a;
    a.img = d.img; a.h0 = d.h0; a.w0 = d.w0; a.rw = d.rw; a.rh = d.rh; a.left = d.left; a.top = d.top; a.side = b.side;
    a.mode = (d.rw == d.w0 && d.rh == d.h0) ? 0 : ((d.w0 == 2 * d.rw && d.h0 == 2 * d.rh) ? 1 : 2);
#pragma unroll
    for (int c = 0; c < 3; ++c) { a.mean[c] = b.mean[c]; a.std[c] = b.std[c]; }
    a.out = b.out + (size_t)blockIdx.y * 3 * b.side * b.side;
    preprocess_pixel(a, blockIdx.x * 256 + threadIdx.x);
}

void launch_preprocess_batch(int n, const unsigned char* const* imgs, const int* geom, int side, const float* mean, const float* stdv,
                             float* out, hipStream_t s)
{
    for (int i0 = 0; i0 < n; i0 += PREP_MAX) {
        const int m = n - i0 < PREP_MAX ? n - i0 : PREP_MAX;
        PrepBatchArgs b{};
        for (int i = 0; i < m; ++i) {
            const int* g = geom + (size_t)(i0 + i) * 6;
            b.im[i] = PrepImg{imgs[i0 + i], g[0], g[1], g[2], g[3], g[4], g[5]};
        }
        b.side = side; b.out = out + (size_t)i0 * 3 * side * side;
        for (int c = 0; c < 3; ++c) { b.mean[c] = mean[c]; b.std[c] = stdv[c]; }
        hipLaunchKernelGGL(preprocess_batch_kernel, dim3((side * side + 255) / 256, m), dim3(256), 0, s, b);
    }
}

void launch_preprocess(const unsigned char* img, int h0, int w0, int rw, int rh, int left, int top, int side,
                       const float* mean, const float* stdv, float* out, hipStream_t s)
{
    PrepArgs a;
    a.img = img; a.h0 = h0; a.w0 = w0; a.rw = rw; a.rh = rh; a.left = left; a.top = top; a.side = side; a.out = out;
    for (int c = 0; c < 3; ++c) { a.mean[c] = mean[c]; a.std[c] = stdv[c]; }
    a.mode = (rw == w0 && rh == h0) ? 0 : ((w0 == 2 * rw && h0 == 2 * rh) ? 1 : 2);
    const int n = side * side;
    hipLaunchKernelGGL(preprocess_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a);
}

}  // namespace ynk

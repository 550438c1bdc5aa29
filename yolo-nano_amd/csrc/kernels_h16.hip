// kernels_h16.hip — the mixed-precision (BASELINE configs[2]: "fp16") training step's kernels for gfx950.
//
// train.py:219-231 runs in fp32; configs[2] names fp16.  This file is the fp16 form of the train-mode network:
//   * activations, pre-BatchNorm conv outputs and every activation gradient are STORED as IEEE fp16 (half the HBM bytes of
//     the fp32 step, which is bandwidth-bound nearly everywhere);
//   * every GEMM-shaped convolution — pointwise 1x1 and dense 3x3, forward, input gradient and weight gradient — runs on
//     v_mfma_f32_32x32x16_f16 with fp32 accumulation;
//   * BatchNorm statistics, parameter gradients, the loss and the optimiser stay fp32 / double on the fp32 master weights;
//   * the loss gradient carries a dynamic loss scale kept on the device (yn_train_h16.inc).
//
// Layout: NHWC with the channel axis PADDED to a multiple of 8 halves, so every row and every channel-half offset is
// 16-byte aligned and one lane moves 8 channels per access.  A ShuffleV2 unit output (2*bf channels, consumed half by half:
// backbone/shufflenetv2.py:70-72) is stored as two planes [x1 | pad][x2 | pad] of bfp = roundup8(bf) channels each
// ("gapped": logical channel c lives at c + (c >= half ? gap : 0)), which keeps the x2 view of bf = 58 / 116 aligned.
// Pad channels hold exact zeros everywhere (every kernel writes whole physical rows), and the packed weights carry zero
// rows / columns at the pad positions, so GEMMs run over physical channel counts with no masking.
#include "yn_internal.h"
#include "yn_h16.h"
#include <stdlib.h>

namespace ynk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ h16x8 ldh8(const h16* p) { return *reinterpret_cast<const h16x8*>(p); }
__device__ __forceinline__ void sth8(h16* p, h16x8 v) { *reinterpret_cast<h16x8*>(p) = v; }
__device__ __forceinline__ h16x8 zero8() { h16x8 z; for (int i = 0; i < 8; ++i) z[i] = (h16)0.0f; return z; }
// value if ok else 0 through an opaque mask: keeps clamped-address loads unconditional (see yn_device.h)
__device__ __forceinline__ h16x8 keep8(h16x8 v, bool ok)
{
    unsigned mk = ok ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(mk));
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 u = __builtin_bit_cast(u32x4, v);
    u.x &= mk; u.y &= mk; u.z &= mk; u.w &= mk;
    return __builtin_bit_cast(h16x8, u);
}
// activation and its derivative without control flow (a run-time `act` in an if-chain compiles to branches per VALUE in the unrolled
// loops: 334 in hbn_apply_kernel); the same bits as v > 0 ? v : (act 1: 0, act 2: 0.1 v, else v)
__device__ __forceinline__ float hact(float v, int act)
{
    const float slope = act == 2 ? 0.1f : 1.0f;
    const unsigned keep = act == 1 ? 0u : 0xffffffffu;
    const float neg = __uint_as_float(__float_as_uint(slope * v) & keep);
    return v > 0.0f ? v : neg;
}
__device__ __forceinline__ float hact_grad(float g, float zz, int act)      // zz > 0 ? g : (act 0: g, act 1: +0, act 2: 0.1 g)
{
    const float slope = act == 2 ? 0.1f : 1.0f;
    const unsigned keep = act == 1 ? 0u : 0xffffffffu;
    const float neg = __uint_as_float(__float_as_uint(slope * g) & keep);
    return zz > 0.0f ? g : neg;
}
__device__ __forceinline__ float hbn_value(float y, float mu, float is, float ga, float be) { return __fmaf_rn(__fmul_rn(__fsub_rn(y, mu), is), ga, be); }
// physical channel p of a gapped row -> logical channel, or -1 for a pad
__device__ __forceinline__ int logical_of(int p, int C, int half, int gap)
{
    if (p < half) return p;
    if (p < half + gap) return -1;
    const int c = p - gap;
    return c < C ? c : -1;
}

// Two channels per instruction: the streaming BatchNorm kernels are instruction-bound at one wavefront per SIMD (16 VALU operations per
// element in the first form: 2.1 ms per 608 / bs-32 step for the backward sums alone), and gfx950 has packed fp32 multiply / add / fma.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pair_of(const h16x8& v, int p) { f32x2 r; r.x = (float)v[2 * p]; r.y = (float)v[2 * p + 1]; return r; }
__device__ __forceinline__ f32x2 splat2(float x) { f32x2 r; r.x = x; r.y = x; return r; }
// xhat and the BatchNorm value with exactly hbn_value's operation sequence (sub, mul, fma): the sign of z decides the activation's branch
// and has to come out the same in the forward apply, the backward sums and the backward apply
__device__ __forceinline__ f32x2 bn_xhat2(f32x2 y, f32x2 mu, f32x2 is) { return (y - mu) * is; }
__device__ __forceinline__ f32x2 bn_value2(f32x2 xh, f32x2 ga, f32x2 be) { return __builtin_elementwise_fma(xh, ga, be); }
// d = g * act'(z): g where z > 0, else g * negslope (0 / 0.1 / 1 for ReLU / LeakyReLU(0.1) / none)
__device__ __forceinline__ f32x2 act_grad2(f32x2 g, f32x2 z, float negslope)
{
    f32x2 m;
    m.x = z.x > 0.0f ? 1.0f : negslope;
    m.y = z.y > 0.0f ? 1.0f : negslope;
    return g * m;
}
__device__ __forceinline__ float act_negslope(int act) { return act == 1 ? 0.0f : (act == 2 ? 0.1f : 1.0f); }

// =================================================================================================
// GEMM-shaped convolutions on the f16 MFMA: pointwise 1x1 (TAPS = 1) and dense 3x3 stride 1 pad 1 (TAPS = 9), used for the
// forward pass and — on transposed (/ flipped) packs — for the input gradients.
//   out[m][n] (+)= sum_tap sum_k A[pix(m, tap)][k] * Wp[tap][k][n] + bias[n]
// A: h16, row m has Kp physical channels at a.in + m*in_ld + in_off (16-byte aligned);  Wp: h16 packed [TAPS][Kp/8][Npad][8]
// (k-octet major: a lane's B fragment = 16 contiguous bytes), Npad a multiple of 32, zero beyond the real rows / columns.
// Block = 4 waves, each owning 32 rows x (32*NT) columns in NT 32x32 f32 accumulators; the block walks K in chunks of 32
// (one A chunk = 128 rows x 4 octets, one B chunk = 4 octets x BN columns, staged through LDS with the next chunk's global
// loads in flight during the MFMAs).  Epilogue through LDS: whole 16-byte row segments leave the block, optionally added to
// what is already there (dX accumulation).  MFMA operand convention: lane l supplies row/column l%32 and the k-octet l/32 of
// a 16-deep step for both A and B, so the k-sum is consistent whatever order the hardware walks the octet in.
// =================================================================================================
template <int NT, int TAPS, int STAT>       // STAT: 0 none, 1 forward statistics of the output, 2 BatchNorm-backward sums of the layer below (HColStat)
__global__ __launch_bounds__(256, NT >= 3 ? 3 : 4) void hgemm_kernel(HGemmArgs a)       // (min waves per SIMD: the statistics epilogues must not cost the main loop its occupancy)
{
    constexpr int BM = 128, BN = 32 * NT, KC = 32, AST = KC + 8;      // A row stride in LDS (halves): 80 bytes, conflict-free b128 reads
    constexpr int OST = BN + 8;                                        // epilogue tile row stride (halves)
    constexpr int LDS_MAIN = BM * AST + (KC / 8) * BN * 8;
    constexpr int LDS_EPI = BM * OST;
    __shared__ __attribute__((aligned(16))) h16 smem[LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI];
    h16* As = smem;
    h16* Bs = smem + BM * AST;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hh = lane >> 5;
    const unsigned gy = (unsigned)(a.Npad / BN), gx8 = gridDim.x / gy;          // XCD-aware decode (see gemm_conv_kernel)
    const unsigned slot = blockIdx.x >> 3;
    const int m0 = (int)((blockIdx.x & 7u) * (gx8 >> 3) + slot / gy) * BM;
    const int n0 = (int)(slot % gy) * BN;
    if (m0 >= a.M) return;
    const int KQ = a.Kp >> 3;                                          // octets per tap
    const int cpt = (a.Kp + KC - 1) / KC;                              // chunks per tap
    const int nchunks = TAPS * cpt;

    // A: thread -> (row = t/4 + 64*i, octet = t%4) of the chunk, i = 0..1
    const int a_oct = t & 3;
    int a_row[2], a_m[2], a_y[2], a_x[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        a_row[i] = (t >> 2) + 64 * i;
        const int m = m0 + a_row[i];
        a_m[i] = m < a.M ? m : -1;
        a_y[i] = a_x[i] = 0;
        if (TAPS == 9 && m < a.M) { const int rem = m % (a.H * a.W); a_y[i] = rem / a.W; a_x[i] = rem - a_y[i] * a.W; }
    }
    constexpr int B_PER = (4 * BN + 255) / 256;                       // B chunk = 4 octets x BN columns = 4*BN 16-byte granules / 256 threads
    // prefetch(): loads only - the zeroing of what a clamped address brought (keep8) happens in stage(), with the flags kept from here.  A mask
    // applied where the value is loaded is a use at the point of issue: hipcc then waits for every load in turn BEFORE the MFMAs of the current
    // chunk, and the K loop runs load -> wait -> MFMA -> barrier in sequence (round 5; the inference GEMMs had the same: yn_device.h).
    h16x8 a_reg[2];
    h16x8 b_reg[B_PER];
    bool a_ok[2], b_ok[B_PER];
    auto prefetch = [&](int c) {
        const int tap = TAPS == 9 ? c / cpt : 0;
        const int kq = (c - tap * cpt) * (KC / 8) + a_oct;             // octet inside the tap
        const int dy = TAPS == 9 ? tap / 3 - 1 : 0, dx = TAPS == 9 ? tap - (tap / 3) * 3 - 1 : 0;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            bool ok = a_m[i] >= 0 && kq < KQ;
            int src = a_m[i] >= 0 ? a_m[i] : 0;
            if (TAPS == 9) {
                const int y = a_y[i] + dy, x = a_x[i] + dx;
                const bool in = y >= 0 && y < a.H && x >= 0 && x < a.W;
                ok = ok && in;
                src = in ? src + dy * a.W + dx : src;                  // clamped to the centre pixel when the tap is outside
            }
            a_reg[i] = ldh8(a.in + (size_t)src * a.in_ld + a.in_off + (kq < KQ ? kq : 0) * 8);
            a_ok[i] = ok;
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int g = t + 256 * i;                                  // granule = (octet o, column n)
            const int o = g / BN, n = g - o * BN;
            const int kqb = (c - tap * cpt) * (KC / 8) + o;
            const bool ok = g < 4 * BN && kqb < KQ;
            b_reg[i] = ldh8(a.Wp + (((size_t)tap * KQ + (ok ? kqb : 0)) * a.Npad + n0 + (g < 4 * BN ? n : 0)) * 8);
            b_ok[i] = ok;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) sth8(As + a_row[i] * AST + a_oct * 8, keep8(a_reg[i], a_ok[i]));
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int g = t + 256 * i;
            if (g < 4 * BN) sth8(Bs + (size_t)g * 8, keep8(b_reg[i], b_ok[i]));
        }
    };
    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    prefetch(0);
    stage();
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) prefetch(c + 1);
        const h16* Ab = As + (wave * 32 + l31) * AST + hh * 8;
        const h16* Bb = Bs + (hh * BN + l31) * 8;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const h16x8 av = ldh8(Ab + ks * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const h16x8 bv = ldh8(Bb + (ks * 2 * BN + nt * 32) * 8);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc[nt], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) {
            __syncthreads();                                            // every wave is done reading the chunk
            stage();
            __syncthreads();
        }
    }
    // ---- epilogue: bias, h16, through LDS so that whole 16-byte row segments are written (and read, when accumulating)
    constexpr int segs = BN / 8;                                        // 16-byte segments per tile row
    constexpr int SEGP = segs <= 4 ? 4 : (segs <= 8 ? 8 : 16), RG = 256 / SEGP, RPT = BM / RG;      // statistics layout: SEGP octet-lanes x RG row groups
    const int st_so = t & (SEGP - 1), st_rg = t / SEGP;
    const int st_pc0 = n0 + st_so * 8;                                   // physical channel of the octet's first column
    const bool st_on = STAT != 0 && st_so < segs && st_pc0 < a.Np;
    h16x8 st_y[STAT == 2 ? RPT : 1];
    if (STAT == 2 && st_on) {                                            // the layer below's pre-BN rows: requested now, needed after the tile has left
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int m = m0 + st_rg + k * RG;
            st_y[k] = ldh8(a.st.y + (size_t)(m < a.M ? m : m0) * a.st.y_ld + st_pc0);
        }
    }
    __syncthreads();
    h16* Os = smem;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ncol = n0 + nt * 32 + l31;
        const float bias = a.bias ? a.bias[ncol] : 0.0f;              // bias has Npad entries (zero padded)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            Os[row * OST + nt * 32 + l31] = (h16)(acc[nt][r] + bias);
        }
    }
    __syncthreads();
    for (int g = t; g < BM * segs; g += 256) {
        const int row = g / segs, sg = g - row * segs;
        const int m = m0 + row, n = n0 + sg * 8;
        if (m >= a.M || n >= a.Np) continue;
        h16x8 v = ldh8(Os + row * OST + sg * 8);
        h16* o = a.out + (size_t)m * a.out_ld + a.out_off + n;
        if (a.accumulate) {
            const h16x8 old = ldh8(o);
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = (h16)((float)v[i] + (float)old[i]);
        }
        sth8(o, v);
    }
    // ---- optional: column sums of the tile just written (HColStat), from its fp16 values in LDS
    if (STAT != 0) {
        __shared__ float st_red[4][16][16];
        const int so = st_so, rg = st_rg, pc0 = st_pc0;
        const bool on = st_on;
        f32x2 s0[4], s1[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) { s0[p] = splat2(0.0f); s1[p] = splat2(0.0f); }
        if (on) {
            if (STAT == 1) {
#pragma unroll 2
                for (int k = 0; k < RPT; ++k) {
                    const int row = rg + k * RG;
                    if (m0 + row >= a.M) continue;
                    const h16x8 v = ldh8(Os + row * OST + so * 8);
#pragma unroll
                    for (int p = 0; p < 4; ++p) { const f32x2 x = pair_of(v, p); s0[p] += x; s1[p] = __builtin_elementwise_fma(x, x, s1[p]); }
                }
            } else {
                const float negslope = act_negslope(a.st.act);
#pragma unroll
                for (int p = 0; p < 4; ++p) {                                    // pair by pair: the coefficients of one pair live at a time (registers decide the main loop's occupancy)
                    const int c0 = logical_of(pc0 + 2 * p, a.st.C, a.st.half, a.st.gap), c1 = logical_of(pc0 + 2 * p + 1, a.st.C, a.st.half, a.st.gap);
                    f32x2 mu, is, ga, be;
                    mu.x = c0 >= 0 ? a.st.mean[c0] : 0.0f; is.x = c0 >= 0 ? a.st.invstd[c0] : 0.0f; ga.x = c0 >= 0 ? a.st.gamma[c0] : 0.0f; be.x = c0 >= 0 ? a.st.beta[c0] : 0.0f;
                    mu.y = c1 >= 0 ? a.st.mean[c1] : 0.0f; is.y = c1 >= 0 ? a.st.invstd[c1] : 0.0f; ga.y = c1 >= 0 ? a.st.gamma[c1] : 0.0f; be.y = c1 >= 0 ? a.st.beta[c1] : 0.0f;
#pragma unroll
                    for (int k = 0; k < RPT; ++k) {
                        const int row = rg + k * RG;
                        const h16x2 v2 = *reinterpret_cast<const h16x2*>(Os + row * OST + so * 8 + 2 * p);
                        f32x2 g; g.x = (float)v2[0]; g.y = (float)v2[1];
                        if (m0 + row >= a.M) g = splat2(0.0f);                   // a row past the end: d = 0
                        const f32x2 xh = bn_xhat2(pair_of(st_y[k], p), mu, is);
                        const f32x2 d = act_grad2(g, bn_value2(xh, ga, be), negslope);
                        s0[p] += d; s1[p] = __builtin_elementwise_fma(d, xh, s1[p]);
                    }
                }
            }
        }
        for (int off = 32; off >= SEGP; off >>= 1) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                s0[p].x += __shfl_xor(s0[p].x, off); s0[p].y += __shfl_xor(s0[p].y, off);
                s1[p].x += __shfl_xor(s1[p].x, off); s1[p].y += __shfl_xor(s1[p].y, off);
            }
        }
        if (lane < SEGP) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                st_red[wave][lane][2 * p] = s0[p].x; st_red[wave][lane][2 * p + 1] = s0[p].y;
                st_red[wave][lane][8 + 2 * p] = s1[p].x; st_red[wave][lane][8 + 2 * p + 1] = s1[p].y;
            }
        }
        __syncthreads();
        if (t < BN && n0 + t < a.Np) {
            const int c = logical_of(n0 + t, a.st.C, a.st.half, a.st.gap);
            if (c >= 0) {
                const int o = t >> 3, j = t & 7;
                double* acc = a.st.acc + (size_t)(blockIdx.x & (HACC_SLOTS - 1)) * 2 * a.st.C;
                atomicAdd(acc + c, (double)((st_red[0][o][j] + st_red[1][o][j]) + (st_red[2][o][j] + st_red[3][o][j])));
                atomicAdd(acc + a.st.C + c, (double)((st_red[0][o][8 + j] + st_red[1][o][8 + j]) + (st_red[2][o][8 + j] + st_red[3][o][8 + j])));
            }
        }
    }
}

template <int NT>
static void launch_hgemm_nt(const HGemmArgs& a, hipStream_t s)
{
    const int BN = 32 * NT;
    const dim3 grid(xcd_grid((unsigned)((a.M + 127) / 128)) * (unsigned)(a.Npad / BN));
    const int stat = !a.st.acc ? 0 : (a.st.y ? 2 : 1);
    if (a.taps == 9) {
        if (stat == 0) hipLaunchKernelGGL((hgemm_kernel<NT, 9, 0>), grid, dim3(256), 0, s, a);
        else if (stat == 1) hipLaunchKernelGGL((hgemm_kernel<NT, 9, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((hgemm_kernel<NT, 9, 2>), grid, dim3(256), 0, s, a);
    } else {
        if (stat == 0) hipLaunchKernelGGL((hgemm_kernel<NT, 1, 0>), grid, dim3(256), 0, s, a);
        else if (stat == 1) hipLaunchKernelGGL((hgemm_kernel<NT, 1, 1>), grid, dim3(256), 0, s, a);
        else hipLaunchKernelGGL((hgemm_kernel<NT, 1, 2>), grid, dim3(256), 0, s, a);
    }
}

// Npad must be a multiple of 32; the widest column tile that divides it is used (<= 128 columns: 64 accumulator registers)
void launch_hgemm(const HGemmArgs& a, hipStream_t s)
{
    const int n32 = a.Npad / 32;
    if (n32 % 4 == 0) launch_hgemm_nt<4>(a, s);
    else if (n32 % 3 == 0) launch_hgemm_nt<3>(a, s);
    else if (n32 % 2 == 0) launch_hgemm_nt<2>(a, s);
    else launch_hgemm_nt<1>(a, s);
}

// =================================================================================================
// Weight gradient of a GEMM-shaped conv on the f16 MFMA:  dW[n][k] = sum_m dY[m][n] * X[pix(m, tap)][k]
// The reduction index m is the MFMA's k, so BOTH operands are needed "m-contiguous per lane", i.e. transposed w.r.t. their
// row-major [m][channel] storage: a block stages MT = 64 rows of dY (<= NB columns) and of X (<= KB columns) row-major in
// LDS with coalesced 16-byte loads and every lane gathers its fragment (8 consecutive m of one channel) with 8 ds_read_u16.
// grid = (column blocks of dY, column blocks of X * taps, M slices); per-slice fp32 copies of dW, reduced — and scattered into
// the reference's weight layout with the channel map of X undone — by hwgrad_reduce_kernel (deterministic, no atomics).
// =================================================================================================
template <int TAPS>
__global__ __launch_bounds__(256) void hwgrad_kernel(HWgradArgs a)
{
    constexpr int MT = 64, NB = 64, KB = 64, ST = 64 + 2;              // LDS row stride (halves): odd dword stride => column gathers spread over banks
    __shared__ __attribute__((aligned(16))) h16 Ds[MT * ST];
    __shared__ __attribute__((aligned(16))) h16 Xs[MT * ST];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hh = lane >> 5;
    const int wn = wave & 1, wk = wave >> 1;                           // wave tile: 32 dY columns x 32 X columns
    const int n0 = blockIdx.x * NB;
    const int kblocks = (a.Kp + KB - 1) / KB;
    const int tap = TAPS == 9 ? blockIdx.y / kblocks : 0;
    const int k0 = (blockIdx.y - tap * kblocks) * KB;
    const int dyy = TAPS == 9 ? tap / 3 - 1 : 0, dxx = TAPS == 9 ? tap - (tap / 3) * 3 - 1 : 0;
    const int slices = gridDim.z;
    const int rows = (((a.M + slices - 1) / slices) + MT - 1) / MT * MT;
    const int m_begin = blockIdx.z * rows, m_end = min(a.M, m_begin + rows);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    // staging: thread -> (row = t/8 + 32*i, octet = t%8), i = 0..1, for both tiles
    const int s_oct = t & 7;
    h16x8 dreg[2], xreg[2];
    bool dok[2], xok[2];                                                // (masks applied in stage(): hgemm_kernel's prefetch)
    auto prefetch = [&](int mt) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int m = mt + (t >> 3) + 32 * i;
            const bool mok = m < m_end;
            const int mc = mok ? m : m_begin;
            const int nn = n0 + s_oct * 8, kk = k0 + s_oct * 8;
            dreg[i] = ldh8(a.dy + (size_t)mc * a.dy_ld + (nn < a.Np ? nn : 0));
            dok[i] = mok && nn < a.Np;
            bool ok = mok && kk < a.Kp;
            int src = mc;
            if (TAPS == 9) {
                const int rem = mc % (a.H * a.W);
                const int y = rem / a.W + dyy, x = rem - (rem / a.W) * a.W + dxx;
                const bool in = y >= 0 && y < a.H && x >= 0 && x < a.W;
                ok = ok && in;
                src = in ? mc + dyy * a.W + dxx : mc;
            }
            xreg[i] = ldh8(a.x + (size_t)src * a.x_ld + a.x_off + (kk < a.Kp ? kk : 0));
            xok[i] = ok;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            dreg[i] = keep8(dreg[i], dok[i]); xreg[i] = keep8(xreg[i], xok[i]);
            const int row = (t >> 3) + 32 * i;
            // ST is not a multiple of 8 halves: store as four 4-byte pieces
            h16x2* d = reinterpret_cast<h16x2*>(Ds + row * ST + s_oct * 8);
            h16x2* x = reinterpret_cast<h16x2*>(Xs + row * ST + s_oct * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                h16x2 dv, xv;
                dv[0] = dreg[i][2 * j]; dv[1] = dreg[i][2 * j + 1];
                xv[0] = xreg[i][2 * j]; xv[1] = xreg[i][2 * j + 1];
                d[j] = dv; x[j] = xv;
            }
        }
    };
    if (m_begin < m_end) {
        prefetch(m_begin);
        for (int mt = m_begin; mt < m_end; mt += MT) {
            __syncthreads();                                            // previous tile fully consumed
            stage();
            __syncthreads();
            if (mt + MT < m_end) prefetch(mt + MT);
#pragma unroll
            for (int ks = 0; ks < MT / 16; ++ks) {
                h16x8 av, bv;
                const int mrow = ks * 16 + hh * 8;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    av[j] = Ds[(mrow + j) * ST + wn * 32 + l31];
                    bv[j] = Xs[(mrow + j) * ST + wk * 32 + l31];
                }
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
            }
        }
    }
    // acc[r]: dY column n = n0 + wn*32 + (r&3) + 8*(r>>2) + 4*hh, X column k = k0 + wk*32 + l31
    float* out = a.partial + (size_t)blockIdx.z * ((size_t)a.Np * a.Kp * TAPS);
    const int k = k0 + wk * 32 + l31;
    if (k < a.Kp) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
            if (n < a.Np) out[((size_t)n * TAPS + tap) * a.Kp + k] = acc[r];
        }
    }
}

// dW (reference layout, logical channels) = sum over slices of partial[s][n][tap][k_phys].  Block = 64 consecutive PACKED entries
// (n, tap, k_phys: coalesced 256-byte reads) x 16 slice lanes — a thread adds every 16th slice with four loads in flight — then an
// LDS tree over the slice lanes; pad entries (k_phys in a gap, n >= N) are dropped, the rest scatter to [n][ci][tap].  Fixed order.
__global__ __launch_bounds__(1024) void hwgrad_reduce_kernel(const float* __restrict__ partial, int slices, int Np, int Kp, int taps,
                                                              int N, int Cin, int half, int gap, float* __restrict__ dw)
{
    __shared__ float red[16][64];
    const int e = threadIdx.x & 63, sl = threadIdx.x >> 6;
    const long total = (long)N * taps * Kp;                             // packed entries of the real output channels
    const long i = (long)blockIdx.x * 64 + e;
    const size_t stride = (size_t)Np * Kp * taps;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    if (i < total) {
        int s = sl;
        for (; s + 48 < slices; s += 64) {
            s0 += partial[(size_t)s * stride + i]; s1 += partial[(size_t)(s + 16) * stride + i];
            s2 += partial[(size_t)(s + 32) * stride + i]; s3 += partial[(size_t)(s + 48) * stride + i];
        }
        for (; s < slices; s += 16) s0 += partial[(size_t)s * stride + i];
    }
    red[sl][e] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (sl == 0 && i < total) {
        float v = 0.0f;
#pragma unroll
        for (int k = 0; k < 16; ++k) v += red[k][e];
        const int kp = (int)(i % Kp);
        const long nt = i / Kp;
        const int tap = (int)(nt % taps), n = (int)(nt / taps);
        int ci = -1;
        if (kp < half) ci = kp; else if (kp >= half + gap) ci = kp - gap;
        if (ci >= 0 && ci < Cin) dw[((size_t)n * Cin + ci) * taps + tap] = v;
    }
}

// -------------------------------------------------------------------------------------------------
// hwgrad_kernel with WIDER wave tiles (round 4): a wavefront owns 32 TN dY columns x 32 TK X columns, so every gathered fragment (eight
// 2-byte LDS reads) feeds TK (TN) MFMAs instead of one - 8 gathers per MFMA at <2, 2> against 16 - and a 128 x 128 block covers a 116- /
// 120-channel layer's whole dW: dY and X are read once per slice, not twice.  Timing experiment that pointed here: the fp16 step without its
// GEMM-shaped weight-gradient launches runs in 6.51 ms against 7.28 - the side stream's 1.9 ms of them cost the critical path 0.77 ms.
// Same partial layout as hwgrad_kernel ([slice][n][tap][k_phys] -> hwgrad_reduce_kernel).
// -------------------------------------------------------------------------------------------------
template <int TAPS, int TN, int TK>
__global__ __launch_bounds__(256) void hwgrad2_kernel(HWgradArgs a)
{
    constexpr int MT = 64, NB = 64 * TN, KB = 64 * TK;
    constexpr int STD = NB + 2, STX = KB + 2;                          // LDS row strides (halves): odd dword strides => column gathers spread over banks
    constexpr int DOCT = NB / 8, XOCT = KB / 8;                        // 16-byte pieces per tile row
    constexpr int DPER = MT * DOCT / 256, XPER = MT * XOCT / 256;      // pieces per thread and tile (2 / 4)
    __shared__ __attribute__((aligned(16))) h16 Ds[MT * STD];
    __shared__ __attribute__((aligned(16))) h16 Xs[MT * STX];
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, hh = lane >> 5;
    const int wn = wave & 1, wk = wave >> 1;
    const int n0 = blockIdx.x * NB;
    const int kblocks = (a.Kp + KB - 1) / KB;
    const int tap = TAPS == 9 ? blockIdx.y / kblocks : 0;
    const int k0 = (blockIdx.y - tap * kblocks) * KB;
    const int dyy = TAPS == 9 ? tap / 3 - 1 : 0, dxx = TAPS == 9 ? tap - (tap / 3) * 3 - 1 : 0;
    const int slices = gridDim.z;
    const int rows = (((a.M + slices - 1) / slices) + MT - 1) / MT * MT;
    const int m_begin = blockIdx.z * rows, m_end = min(a.M, m_begin + rows);
    // column tiles of this wavefront that hold real columns (a 96-channel layer under a 128-wide block: the upper tiles are skipped)
    bool ton[TN], tok[TK];
#pragma unroll
    for (int i = 0; i < TN; ++i) ton[i] = n0 + (wn * TN + i) * 32 < a.Np;
#pragma unroll
    for (int i = 0; i < TK; ++i) tok[i] = k0 + (wk * TK + i) * 32 < a.Kp;
    f32x16 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    h16x8 dreg[DPER], xreg[XPER];
    bool dok[DPER], xok[XPER];                                          // (masks applied in stage(): hgemm_kernel's prefetch)
    auto prefetch = [&](int mt) {
#pragma unroll
        for (int i = 0; i < DPER; ++i) {
            const int g = t + 256 * i;
            const int m = mt + g / DOCT, nn = n0 + (g % DOCT) * 8;
            const bool ok = m < m_end && nn < a.Np;
            dreg[i] = ldh8(a.dy + (size_t)(m < m_end ? m : m_begin) * a.dy_ld + (nn < a.Np ? nn : 0));
            dok[i] = ok;
        }
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            const int g = t + 256 * i;
            const int m = mt + g / XOCT, kk = k0 + (g % XOCT) * 8;
            const int mc = m < m_end ? m : m_begin;
            bool ok = m < m_end && kk < a.Kp;
            int src = mc;
            if (TAPS == 9) {
                const int rem = mc % (a.H * a.W);
                const int y = rem / a.W + dyy, x = rem - (rem / a.W) * a.W + dxx;
                const bool in = y >= 0 && y < a.H && x >= 0 && x < a.W;
                ok = ok && in;
                src = in ? mc + dyy * a.W + dxx : mc;
            }
            xreg[i] = ldh8(a.x + (size_t)src * a.x_ld + a.x_off + (kk < a.Kp ? kk : 0));
            xok[i] = ok;
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < DPER; ++i) dreg[i] = keep8(dreg[i], dok[i]);
#pragma unroll
        for (int i = 0; i < XPER; ++i) xreg[i] = keep8(xreg[i], xok[i]);
#pragma unroll
        for (int i = 0; i < DPER; ++i) {
            const int g = t + 256 * i;
            h16x2* d = reinterpret_cast<h16x2*>(Ds + (g / DOCT) * STD + (g % DOCT) * 8);      // the stride is not a multiple of 8 halves: four 4-byte pieces
#pragma unroll
            for (int j = 0; j < 4; ++j) { h16x2 v; v[0] = dreg[i][2 * j]; v[1] = dreg[i][2 * j + 1]; d[j] = v; }
        }
#pragma unroll
        for (int i = 0; i < XPER; ++i) {
            const int g = t + 256 * i;
            h16x2* x = reinterpret_cast<h16x2*>(Xs + (g / XOCT) * STX + (g % XOCT) * 8);
#pragma unroll
            for (int j = 0; j < 4; ++j) { h16x2 v; v[0] = xreg[i][2 * j]; v[1] = xreg[i][2 * j + 1]; x[j] = v; }
        }
    };
    if (m_begin < m_end) {
        prefetch(m_begin);
        for (int mt = m_begin; mt < m_end; mt += MT) {
            __syncthreads();                                            // previous tile fully consumed
            stage();
            __syncthreads();
            if (mt + MT < m_end) prefetch(mt + MT);
#pragma unroll
            for (int ks = 0; ks < MT / 16; ++ks) {
                const int mrow = ks * 16 + hh * 8;
                h16x8 av[TN], bv[TK];
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) av[i][j] = Ds[(mrow + j) * STD + (wn * TN + i) * 32 + l31];
#pragma unroll
                for (int i = 0; i < TK; ++i)
#pragma unroll
                    for (int j = 0; j < 8; ++j) bv[i][j] = Xs[(mrow + j) * STX + (wk * TK + i) * 32 + l31];
#pragma unroll
                for (int i = 0; i < TN; ++i)
#pragma unroll
                    for (int j = 0; j < TK; ++j)
                        if (ton[i] && tok[j]) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[i], bv[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    // acc[i][j][r]: dY column n = n0 + (wn TN + i) 32 + (r&3) + 8 (r>>2) + 4 hh, X column k = k0 + (wk TK + j) 32 + l31
    float* out = a.partial + (size_t)blockIdx.z * ((size_t)a.Np * a.Kp * TAPS);
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int k = k0 + (wk * TK + j) * 32 + l31;
            if (k >= a.Kp) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (wn * TN + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                if (n < a.Np) out[((size_t)n * TAPS + tap) * a.Kp + k] = acc[i][j][r];
            }
        }
}

void launch_hwgrad(const HWgradArgs& a, hipStream_t s)
{
    // wider wave tiles where the layer has the columns for them (YN_WG_WIDE=0: the 64 x 64 blocks everywhere)
    static const int wide = getenv("YN_WG_WIDE") ? atoi(getenv("YN_WG_WIDE")) : 1;
    const int TN = (wide && a.Np > 64) ? 2 : 1, TK = (wide && a.Kp > 64) ? 2 : 1;
    const int gn = (a.Np + 64 * TN - 1) / (64 * TN), gk = (a.Kp + 64 * TK - 1) / (64 * TK) * a.taps;
    static const int wg_blocks = getenv("YN_WG_BLOCKS") ? atoi(getenv("YN_WG_BLOCKS")) : 2048;
    int slices = wg_blocks / (gn * gk);
    if (slices > 512) slices = 512;
    static const int slice_rows = getenv("YN_WG_SLICE_ROWS") ? atoi(getenv("YN_WG_SLICE_ROWS")) : 512;
    const int max_slices = (a.M + slice_rows - 1) / slice_rows;
    if (slices > max_slices) slices = max_slices;
    const long nk = (long)a.Np * a.Kp * a.taps;
    if ((long)slices * nk > (long)a.partial_cap) slices = (int)((long)a.partial_cap / nk);
    if (slices < 1) slices = 1;
    const dim3 grid(gn, gk, slices);
#define YN_WG2(taps_, tn, tk) hipLaunchKernelGGL((hwgrad2_kernel<taps_, tn, tk>), grid, dim3(256), 0, s, a)
    if (TN == 2 && TK == 2) { if (a.taps == 9) YN_WG2(9, 2, 2); else YN_WG2(1, 2, 2); }
    else if (TN == 2) { if (a.taps == 9) YN_WG2(9, 2, 1); else YN_WG2(1, 2, 1); }
    else if (TK == 2) { if (a.taps == 9) YN_WG2(9, 1, 2); else YN_WG2(1, 1, 2); }
    else if (a.taps == 9) hipLaunchKernelGGL(hwgrad_kernel<9>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(hwgrad_kernel<1>, grid, dim3(256), 0, s, a);
#undef YN_WG2
    const long total = (long)a.N * a.taps * a.Kp;
    hipLaunchKernelGGL(hwgrad_reduce_kernel, dim3((unsigned)((total + 63) / 64)), dim3(1024), 0, s, a.partial, slices, a.Np, a.Kp, a.taps,
                       a.N, a.Cin, a.half, a.gap, a.dw);
}

// =================================================================================================
// Depthwise 3x3 (stride 1 / 2, pad 1), h16 in / out, fp32 taps [9][Cp] (+ bias [Cp] or null), fp32 math.
// thread = one channel octet of one output pixel; XCD-contiguous pixel order (the 3 input rows of a window share an L2).
// Serves the forward convs and — with flipped taps — the stride-1 input gradient (accumulate: dX += ).
// =================================================================================================
template <int STRIDE>
__global__ __launch_bounds__(256) void hdw_kernel(HDwArgs a)
{
    const int Ho = (a.H - 1) / STRIDE + 1, Wo = (a.W - 1) / STRIDE + 1;
    const int OC = a.Cp >> 3;
    const long total = (long)a.B * Ho * Wo * OC;
    const long i = (long)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (i >= total) return;
    const int oc = (int)(i % OC);
    long p = i / OC;
    const int ox = (int)(p % Wo); const long q = p / Wo;
    const int oy = (int)(q % Ho), b = (int)(q / Ho);
    const int c = oc * 8;
    h16x8 v[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * STRIDE - 1 + ky;
        const bool yok = iy >= 0 && iy < a.H;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * STRIDE - 1 + kx;
            const bool ok = yok && ix >= 0 && ix < a.W;
            const size_t src = ((size_t)(b * a.H + (yok ? iy : 0)) * a.W + (ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix)));
            v[ky * 3 + kx] = keep8(ldh8(a.in + src * a.in_ld + a.in_off + c), ok);
        }
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = a.bias ? a.bias[c + j] : 0.0f;
#pragma unroll
    for (int tp = 0; tp < 9; ++tp) {
        const float4 w0 = *reinterpret_cast<const float4*>(a.w + (size_t)tp * a.Cp + c);
        const float4 w1 = *reinterpret_cast<const float4*>(a.w + (size_t)tp * a.Cp + c + 4);
        acc[0] = __builtin_fmaf((float)v[tp][0], w0.x, acc[0]); acc[1] = __builtin_fmaf((float)v[tp][1], w0.y, acc[1]);
        acc[2] = __builtin_fmaf((float)v[tp][2], w0.z, acc[2]); acc[3] = __builtin_fmaf((float)v[tp][3], w0.w, acc[3]);
        acc[4] = __builtin_fmaf((float)v[tp][4], w1.x, acc[4]); acc[5] = __builtin_fmaf((float)v[tp][5], w1.y, acc[5]);
        acc[6] = __builtin_fmaf((float)v[tp][6], w1.z, acc[6]); acc[7] = __builtin_fmaf((float)v[tp][7], w1.w, acc[7]);
    }
    h16* o = a.out + (size_t)p * a.out_ld + a.out_off + c;
    h16x8 r;
    if (a.accumulate) {
        const h16x8 old = ldh8(o);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (h16)(acc[j] + (float)old[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (h16)acc[j];
    }
    sth8(o, r);
}

// -------------------------------------------------------------------------------------------------
// Stride-1 depthwise 3x3 as RUNS (round 4): thread = one channel octet of R = 4 consecutive output pixels of a row.  hdw_kernel<1>
// loads nine 16-byte vectors per output (9x the tensor through L1 / L2: 20 us for an 11 MB stage-3 map, twice the BatchNorm apply pass
// beside it); a run shares its 3 x 6 window: 4.5 loads per output.  Same fma chain per output (taps 0..8 onto the bias): same bits.
// A workgroup is PB = 256 / OC runs x OC octets and walks NR blocks of runs, so the launch has a few hundred workgroups and the optional
// statistics cost 2 C double atomics each (HColStat, as in hgemm_kernel's epilogue - the separate hcol_reduce launch and its pass over
// the tensor disappear):
//   STAT 1 (forward):        acc[0][c] += sum out, acc[1][c] += sum out^2 of the fp16 values just stored - the BatchNorm statistics;
//   STAT 2 (input gradient): the output is dz of the layer BELOW (its BN + activation fed this conv); with that layer's pre-BN output y:
//                            acc[0][c] += sum d, acc[1][c] += sum d * xhat, d = dz * act'(BN(y))     (hcol_reduce_kernel<2>'s sums).
// -------------------------------------------------------------------------------------------------
template <int STAT>
__global__ __launch_bounds__(256) void hdw_run_kernel(HDwArgs a, int NR)
{
    constexpr int R = 4;
    const int OC = a.Cp >> 3, PB = 256 / OC;
    const int RW = (a.W + R - 1) / R;
    const long runs = (long)a.B * a.H * RW;
    const int t = threadIdx.x;
    const int oc = t % OC, rl = t / OC, c = oc * 8;
    const bool worker = rl < PB;
    f32x2 s0[4], s1[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { s0[p] = splat2(0.0f); s1[p] = splat2(0.0f); }
    f32x2 mu[4], is[4], ga[4], be[4];
    float negslope = 1.0f;
    if (STAT == 2) {
        negslope = act_negslope(a.st.act);
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int c0 = logical_of(c + 2 * p, a.st.C, a.st.half, a.st.gap), c1 = logical_of(c + 2 * p + 1, a.st.C, a.st.half, a.st.gap);
            mu[p].x = c0 >= 0 ? a.st.mean[c0] : 0.0f; is[p].x = c0 >= 0 ? a.st.invstd[c0] : 0.0f; ga[p].x = c0 >= 0 ? a.st.gamma[c0] : 0.0f; be[p].x = c0 >= 0 ? a.st.beta[c0] : 0.0f;
            mu[p].y = c1 >= 0 ? a.st.mean[c1] : 0.0f; is[p].y = c1 >= 0 ? a.st.invstd[c1] : 0.0f; ga[p].y = c1 >= 0 ? a.st.gamma[c1] : 0.0f; be[p].y = c1 >= 0 ? a.st.beta[c1] : 0.0f;
        }
    }
    float bias[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) bias[j] = a.bias ? a.bias[c + j] : 0.0f;
    const long blk = (long)xcd_block(blockIdx.x, gridDim.x);
    for (int n = 0; n < NR; ++n) {
        const long run = (blk * NR + n) * PB + rl;
        if (!worker || run >= runs) continue;
        const int xr = (int)(run % RW); const long q = run / RW;
        const int oy = (int)(q % a.H), b = (int)(q / a.H);
        const int x0 = xr * R;
        h16x8 v[3][R + 2];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy - 1 + ky;
            const bool yok = iy >= 0 && iy < a.H;
            const h16* rowp = a.in + ((size_t)(b * a.H + (yok ? iy : 0)) * a.W) * a.in_ld + a.in_off + c;
#pragma unroll
            for (int i = 0; i < R + 2; ++i) {
                const int ix = x0 - 1 + i;
                const bool ok = yok && ix >= 0 && ix < a.W;
                v[ky][i] = keep8(ldh8(rowp + (size_t)(ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix)) * a.in_ld), ok);
            }
        }
        const size_t p0 = ((size_t)(b * a.H + oy)) * a.W + x0;            // first output pixel of the run
        h16x8 yv[STAT == 2 ? R : 1], old[R];
        if (STAT == 2) {
#pragma unroll
            for (int o = 0; o < R; ++o) yv[o] = ldh8(a.st.y + (p0 + (x0 + o < a.W ? o : 0)) * a.st.y_ld + c);
        }
        if (a.accumulate) {
#pragma unroll
            for (int o = 0; o < R; ++o) old[o] = ldh8(a.out + (p0 + (x0 + o < a.W ? o : 0)) * a.out_ld + a.out_off + c);
        }
        float acc[R][8];
#pragma unroll
        for (int o = 0; o < R; ++o)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[o][j] = bias[j];
#pragma unroll
        for (int tp = 0; tp < 9; ++tp) {
            const float4 w0 = *reinterpret_cast<const float4*>(a.w + (size_t)tp * a.Cp + c);
            const float4 w1 = *reinterpret_cast<const float4*>(a.w + (size_t)tp * a.Cp + c + 4);
            const float w8[8] = {w0.x, w0.y, w0.z, w0.w, w1.x, w1.y, w1.z, w1.w};
#pragma unroll
            for (int o = 0; o < R; ++o)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[o][j] = __builtin_fmaf((float)v[tp / 3][o + tp % 3][j], w8[j], acc[o][j]);
        }
#pragma unroll
        for (int o = 0; o < R; ++o) {
            if (x0 + o >= a.W) continue;
            h16x8 r;
            if (a.accumulate) {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (h16)(acc[o][j] + (float)old[o][j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) r[j] = (h16)acc[o][j];
            }
            sth8(a.out + (p0 + o) * a.out_ld + a.out_off + c, r);
            if (STAT == 1) {
#pragma unroll
                for (int p = 0; p < 4; ++p) { const f32x2 x = pair_of(r, p); s0[p] += x; s1[p] = __builtin_elementwise_fma(x, x, s1[p]); }
            } else if (STAT == 2) {
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x2 xh = bn_xhat2(pair_of(yv[o], p), mu[p], is[p]);
                    const f32x2 d = act_grad2(pair_of(r, p), bn_value2(xh, ga[p], be[p]), negslope);
                    s0[p] += d; s1[p] = __builtin_elementwise_fma(d, xh, s1[p]);
                }
            }
        }
    }
    if (STAT != 0) {
        // the workgroup's runs -> one value per (sum, channel): [PB][OC * 16] floats through LDS, then 2 C double atomics into this block's slot
        __shared__ float red[256 * 16];
        if (worker) {
            float* rp = red + (rl * OC + oc) * 16;
#pragma unroll
            for (int p = 0; p < 4; ++p) { rp[2 * p] = s0[p].x; rp[2 * p + 1] = s0[p].y; rp[8 + 2 * p] = s1[p].x; rp[8 + 2 * p + 1] = s1[p].y; }
        }
        __syncthreads();
        for (int e = t; e < OC * 16; e += 256) {                         // e = (octet, which sum, channel in octet)
            const int o8 = e >> 4, k = e & 15;
            const int lc = logical_of(o8 * 8 + (k & 7), a.st.C, a.st.half, a.st.gap);
            if (lc < 0) continue;
            float s = 0.0f;
            for (int r = 0; r < PB; ++r) s += red[(r * OC + o8) * 16 + k];
            double* acc = a.st.acc + (size_t)(blockIdx.x & (HACC_SLOTS - 1)) * 2 * a.st.C;
            atomicAdd(acc + (k >> 3) * a.st.C + lc, (double)s);
        }
    }
}

void launch_hdw(const HDwArgs& a, hipStream_t s)
{
    const int Ho = (a.H - 1) / a.stride + 1, Wo = (a.W - 1) / a.stride + 1;
    const long total = (long)a.B * Ho * Wo * (a.Cp >> 3);
    const dim3 grid(xcd_grid((unsigned)((total + 255) / 256)));
    static const int runs_on = getenv("YN_HDW_RUNS") ? atoi(getenv("YN_HDW_RUNS")) : 1;        // 0: one output per thread (A/B runs; no statistics then)
    if (a.stride == 1 && (runs_on || a.st.acc) && a.Cp <= 256) {
        const int OC = a.Cp >> 3, PB = 256 / OC;
        const long runs = (long)a.B * a.H * ((a.W + 3) / 4);
        const long nb1 = (runs + PB - 1) / PB;                              // workgroups at one block of runs each
        static const int gtarget = getenv("YN_HDW_G") ? atoi(getenv("YN_HDW_G")) : 256;      // with statistics: one workgroup per CU - 2 C double atomics each (7.56 / 7.62 / 7.66 / 7.72 / 7.77 ms per step at 256 / 512 / 1 024 / 2 048 / 4 096)
        const long cap = a.st.acc ? gtarget : 4096;
        const int NR = (int)((nb1 + cap - 1) / cap);
        const dim3 g2(xcd_grid((unsigned)((nb1 + NR - 1) / NR)));
        const int stat = !a.st.acc ? 0 : (a.st.y ? 2 : 1);
        if (stat == 0) hipLaunchKernelGGL(hdw_run_kernel<0>, g2, dim3(256), 0, s, a, NR);
        else if (stat == 1) hipLaunchKernelGGL(hdw_run_kernel<1>, g2, dim3(256), 0, s, a, NR);
        else hipLaunchKernelGGL(hdw_run_kernel<2>, g2, dim3(256), 0, s, a, NR);
        return;
    }
    if (a.stride == 1) hipLaunchKernelGGL(hdw_kernel<1>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(hdw_kernel<2>, grid, dim3(256), 0, s, a);
}

// depthwise 3x3 stride-2 input gradient (gather form, see dw_dgrad_s2_kernel): thread = (input pixel, channel octet); w = forward taps [9][Cp]
__global__ __launch_bounds__(256) void hdw_dgrad_s2_kernel(const h16* __restrict__ dy, int dy_ld, const float* __restrict__ w, int B, int H, int W, int Cp,
                                                            h16* __restrict__ dx, int dx_ld, int dx_off, int accumulate)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int OC = Cp >> 3;
    const long total = (long)B * H * W * OC;
    const long i = (long)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (i >= total) return;
    const int oc = (int)(i % OC);
    long p = i / OC;
    const int ix = (int)(p % W); const long q = p / W;
    const int iy = (int)(q % H), b = (int)(q / H);
    const int c = oc * 8;
    int oy[2], ky[2], ox[2], kx[2];
    bool vy[2], vx[2];
    oy[0] = (iy + 1) >> 1; ky[0] = iy + 1 - 2 * oy[0]; vy[0] = oy[0] < Ho;
    oy[1] = oy[0] - 1;     ky[1] = 2;                  vy[1] = ky[0] == 0 && oy[1] >= 0;
    ox[0] = (ix + 1) >> 1; kx[0] = ix + 1 - 2 * ox[0]; vx[0] = ox[0] < Wo;
    ox[1] = ox[0] - 1;     kx[1] = 2;                  vx[1] = kx[0] == 0 && ox[1] >= 0;
    h16x8 g[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = u >> 1, f = u & 1;
        const bool ok = vy[e] && vx[f];
        g[u] = keep8(ldh8(dy + ((size_t)(b * Ho + (vy[e] ? oy[e] : 0)) * Wo + (vx[f] ? ox[f] : 0)) * dy_ld + c), ok);
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const float* wp = w + (size_t)(ky[u >> 1] * 3 + kx[u & 1]) * Cp + c;
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[j] += (float)g[u][j] * wp[j];
    }
    h16* o = dx + (size_t)p * dx_ld + dx_off + c;
    h16x8 r;
    if (accumulate) { const h16x8 old = ldh8(o);
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (h16)(acc[j] + (float)old[j]);
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (h16)acc[j];
    }
    sth8(o, r);
}

void launch_hdw_dgrad_s2(const h16* dy, int dy_ld, const float* w, int B, int H, int W, int Cp, h16* dx, int dx_ld, int dx_off, int accumulate, hipStream_t s)
{
    const long total = (long)B * H * W * (Cp >> 3);
    hipLaunchKernelGGL(hdw_dgrad_s2_kernel, dim3(xcd_grid((unsigned)((total + 255) / 256))), dim3(256), 0, s, dy, dy_ld, w, B, H, W, Cp, dx, dx_ld, dx_off, accumulate);
}

// =================================================================================================
// Column reductions over an [M][Cp] h16 matrix.  Block = OL octet-lanes x (256/OL) row-lanes; a thread owns 8 channels and
// walks down the rows with four independent 16-byte loads in flight (fp32 batch partials, two channels per instruction, entering double
// accumulators once per batch); row-lanes are combined through
// LDS and ONE double atomic per channel per block goes into the block's accumulator slot (HACC_SLOTS copies).
//   MODE 0  stats:     acc[0][c] += sum y,   acc[1][c] += sum y*y
//   MODE 2  BN bwd:    acc[0][c] += sum dyh, acc[1][c] += sum dyh * xhat      (dyh = dz * act'(BN(y)), xhat = (y - mean) * invstd)
//   MODE 3  column sum of y into the fp32 gradient slots (bias gradient)
// Channel index in acc / mean / gamma is the LOGICAL channel (logical_of); pad channels are skipped.
// dz: dense with y's own map (dz_odd = 0), or the odd logical channels of a gapped 2C-channel tensor (dz_odd = 1: channel c
// of this layer is logical channel 2c+1 there — the concat+shuffle of backbone/shufflenetv2.py:72-74).
// =================================================================================================
// dz_odd addressing of one octet of this layer's channels (c = p0 .. p0+7 <-> logical 2c+1 of the gapped 2C-channel unit gradient): the 16
// interleaved values are the logical positions l0 = 2 p0 .. l0 + 15, i.e. two 8-half vectors d0 / d1, each addressed in the plane of its
// first element; when the plane boundary `half` falls INSIDE one of them (bf = 116: the octet of c = 56..63) that vector's tail comes
// from a third load `e` in second-plane addressing.  Every lane thus gets its row in at most three 16-byte loads (the first version sent
// the straddling octet and the ragged last one through 8 scalar loads per row, issued where they were needed: a full memory latency per
// row for every wavefront holding such a lane — the reduce over a 10 MB stage-3 tensor took 23 us against 9 us for the statistics pass).
struct DzOdd { int o0, o1, oe, sv, cut; bool load1, vec; };
__device__ __forceinline__ DzOdd dz_odd_map_g(int C, int dz_ld, int half, int gap, int p0)
{
    DzOdd m;
    const int l0 = 2 * p0;
    m.o0 = l0 + (l0 >= half ? gap : 0);
    m.o1 = l0 + 8 + (l0 + 8 >= half ? gap : 0);
    m.sv = (l0 < half && half < l0 + 8) ? 0 : ((l0 + 8 < half && half < l0 + 16) ? 1 : -1);
    m.oe = m.sv >= 0 ? l0 + 8 * m.sv + gap : m.o0;
    m.cut = m.sv >= 0 ? half - (l0 + 8 * m.sv) : 8;
    m.load1 = p0 + 4 < C;
    m.vec = m.o0 + 8 <= dz_ld && (!m.load1 || m.o1 + 8 <= dz_ld) && m.oe + 8 <= dz_ld;
    if (!m.load1) m.o1 = m.o0;                              // pad channels only: any valid address, the values are never used
    return m;
}
__device__ __forceinline__ DzOdd dz_odd_map(const HRedArgs& a, int p0) { return dz_odd_map_g(a.C, a.dz_ld, a.dz_half, a.dz_gap, p0); }
__device__ __forceinline__ void dz_odd_pick(const DzOdd& m, const h16x8& d0, const h16x8& d1, const h16x8& e, float (&g)[8])
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int i = 2 * j + 1;
        g[j] = (float)((m.sv == 0 && i >= m.cut) ? e[i] : d0[i]);
        g[4 + j] = m.load1 ? (float)((m.sv == 1 && i >= m.cut) ? e[i] : d1[i]) : 0.0f;
    }
}

__device__ __forceinline__ void load_dz8(const HRedArgs& a, size_t row, int p0, float (&g)[8])
{
    if (!a.dz_odd) {
        const h16x8 v = ldh8(a.dz + row * a.dz_ld + a.dz_off + p0);
#pragma unroll
        for (int j = 0; j < 8; ++j) g[j] = (float)v[j];
    } else {
        const DzOdd m = dz_odd_map(a, p0);                              // y is dense here (half == C): physical == logical
        if (m.vec) {
            const h16* q = a.dz + row * a.dz_ld;
            const h16x8 d0 = ldh8(q + m.o0), d1 = ldh8(q + m.o1), e = ldh8(q + m.oe);
            dz_odd_pick(m, d0, d1, e, g);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = p0 + j;
                const int l = 2 * c + 1;
                const int pp = l + (l >= a.dz_half ? a.dz_gap : 0);
                g[j] = c < a.C ? (float)a.dz[row * a.dz_ld + pp] : 0.0f;
            }
        }
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void hcol_reduce_kernel(HRedArgs a)
{
    const int OL = a.lanes, rowsPer = 256 / OL;                 // OL <= 32 (Cp <= 256): every octet has at least two row-lanes per wave
    const int ol = threadIdx.x & (OL - 1), rl = threadIdx.x / OL;
    const int OC = a.Cp >> 3;
    const bool live = ol < OC;
    const int p0 = ol * 8;
    double s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.0; s1[j] = 0.0; }
    int lc[8];
    f32x2 mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) lc[j] = live ? logical_of(p0 + j, a.C, a.half, a.gap) : -1;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        mu[p] = is[p] = ga[p] = be[p] = splat2(0.0f);
        if (MODE == 2) {
            if (lc[2 * p] >= 0) { mu[p].x = a.mean[lc[2 * p]]; is[p].x = a.invstd[lc[2 * p]]; ga[p].x = a.gamma[lc[2 * p]]; be[p].x = a.beta[lc[2 * p]]; }
            if (lc[2 * p + 1] >= 0) { mu[p].y = a.mean[lc[2 * p + 1]]; is[p].y = a.invstd[lc[2 * p + 1]]; ga[p].y = a.gamma[lc[2 * p + 1]]; be[p].y = a.beta[lc[2 * p + 1]]; }
        }
    }
    const float negslope = act_negslope(a.act);
    if (live) {
        // Software pipeline: the next batch of U rows is requested before the current one is reduced (one workgroup per CU means one
        // wavefront per SIMD: nothing else hides a batch's load latency).  A batch is summed in fp32, two channels per instruction, and
        // enters the double accumulators once (4 rows: the partial carries 2-3 ulps of fp32 at most).
        constexpr int U = 4;
        const long step = (long)gridDim.x * rowsPer;
        DzOdd dm{};
        if (MODE == 2 && a.dz_odd) dm = dz_odd_map(a, p0);
        const bool slowdz = MODE == 2 && a.dz_odd && !dm.vec;
        const size_t dzo = (MODE == 2 && a.dz_odd) ? (size_t)dm.o0 : (size_t)(a.dz_off + p0);
        struct Batch { h16x8 v[U], d0[U], d1[U], e[U]; bool ok[U]; };      // two of them, indexed statically (a run-time index spills to scratch)
        auto issue = [&](Batch& q, long r) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                q.ok[u] = r + u * step < a.M;
                const size_t row = (size_t)(q.ok[u] ? r + u * step : (r < a.M ? r : a.M - 1));
                q.v[u] = ldh8(a.y + row * a.y_ld + a.y_off + p0);
                if (MODE == 2 && !slowdz) {
                    q.d0[u] = ldh8(a.dz + row * a.dz_ld + dzo);
                    if (a.dz_odd) { q.d1[u] = ldh8(a.dz + row * a.dz_ld + dm.o1); q.e[u] = ldh8(a.dz + row * a.dz_ld + dm.oe); }
                }
            }
        };
        auto reduce = [&](const Batch& q, long r) {
            f32x2 t0[4], t1[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) { t0[p] = splat2(0.0f); t1[p] = splat2(0.0f); }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                h16x8 g = zero8();
                if (MODE == 2) {
                    if (a.dz_odd) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const int i = 2 * j + 1;
                            g[j] = (dm.sv == 0 && i >= dm.cut) ? q.e[u][i] : q.d0[u][i];
                            g[4 + j] = dm.load1 ? ((dm.sv == 1 && i >= dm.cut) ? q.e[u][i] : q.d1[u][i]) : (h16)0.0f;
                        }
                    } else g = q.d0[u];
                    g = keep8(g, q.ok[u]);                       // a row past the end contributes d = 0 to both sums
                }
                const h16x8 v = (MODE == 2) ? q.v[u] : keep8(q.v[u], q.ok[u]);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x2 yv = pair_of(v, p);
                    if (MODE == 0) { t0[p] += yv; t1[p] = __builtin_elementwise_fma(yv, yv, t1[p]); }
                    else if (MODE == 3) { t0[p] += yv; }
                    else {
                        const f32x2 xh = bn_xhat2(yv, mu[p], is[p]);
                        const f32x2 d = act_grad2(pair_of(g, p), bn_value2(xh, ga[p], be[p]), negslope);
                        t0[p] += d;
                        t1[p] = __builtin_elementwise_fma(d, xh, t1[p]);
                    }
                }
            }
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                s0[2 * p] += (double)t0[p].x; s0[2 * p + 1] += (double)t0[p].y;
                if (MODE != 3) { s1[2 * p] += (double)t1[p].x; s1[2 * p + 1] += (double)t1[p].y; }
            }
        };
        Batch A, B;
        const long bstep = U * step;
        long r = (long)blockIdx.x * rowsPer + rl;
        if (slowdz) {                                           // ragged last octet (C = 5..7 mod 8: no layer of the networks): one row at a time
            for (; r < a.M; r += step) {
                float gf[8];
                load_dz8(a, (size_t)r, p0, gf);
                const h16x8 v = ldh8(a.y + (size_t)r * a.y_ld + a.y_off + p0);
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const f32x2 xh = bn_xhat2(pair_of(v, p), mu[p], is[p]);
                    f32x2 gp; gp.x = gf[2 * p]; gp.y = gf[2 * p + 1];
                    const f32x2 d = act_grad2(gp, bn_value2(xh, ga[p], be[p]), negslope);
                    s0[2 * p] += (double)d.x; s0[2 * p + 1] += (double)d.y;
                    s1[2 * p] += (double)(d.x * xh.x); s1[2 * p + 1] += (double)(d.y * xh.y);
                }
            }
            r = a.M;
        }
        if (r < a.M) issue(A, r);
        while (r < a.M) {
            if (r + bstep < a.M) issue(B, r + bstep);
            reduce(A, r);
            r += bstep;
            if (r >= a.M) break;
            if (r + bstep < a.M) issue(A, r + bstep);
            reduce(B, r);
            r += bstep;
        }
    }
    // combine the row-lanes: inside a wave by xor-shuffles over the lane bits above the octet-lane bits (the row-lanes of one octet
    // sit OL lanes apart), then the four waves through LDS; one thread per octet issues the atomics
    __shared__ double red[4][32][16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (OL <= 32) {
        for (int off = 32; off >= OL; off >>= 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { s0[j] += __shfl_xor(s0[j], off); s1[j] += __shfl_xor(s1[j], off); }
        }
        if (lane < OL) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { red[wave][lane][j] = s0[j]; red[wave][lane][8 + j] = s1[j]; }
        }
    }
    __syncthreads();
    if (threadIdx.x < OL && live) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s0[j] = (red[0][ol][j] + red[1][ol][j]) + (red[2][ol][j] + red[3][ol][j]);
            s1[j] = (red[0][ol][8 + j] + red[1][ol][8 + j]) + (red[2][ol][8 + j] + red[3][ol][8 + j]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (lc[j] < 0) continue;
            if (MODE == 3) {
                atomicAdd(a.facc + (size_t)(blockIdx.x & (GRAD_SLOTS - 1)) * a.slot_stride + lc[j], (float)s0[j]);
            } else {
                double* acc = a.acc + (size_t)(blockIdx.x & (HACC_SLOTS - 1)) * 2 * a.C;
                atomicAdd(acc + lc[j], s0[j]);
                atomicAdd(acc + a.C + lc[j], s1[j]);
            }
        }
    }
}

static int hlanes_for(int Cp) { int l = 1; while (l < (Cp >> 3) && l < 256) l <<= 1; return l; }
static int hreduce_blocks(long M, int rowsPer, int cap)
{
    long b = (M + (long)rowsPer * 4 - 1) / ((long)rowsPer * 4);
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

void launch_hcol_reduce(const HRedArgs& a0, int mode, hipStream_t s)
{
    HRedArgs a = a0;
    a.lanes = hlanes_for(a.Cp);
    // one block per CU measured best (11.9 ms per 608 / bs-32 step against 12.0 at 512 and 12.4 at 1024 blocks): the per-block tail
    // (cross-wave combine + 2*C fp64 atomics) outweighs the extra loads in flight
    static const int gmax = getenv("YN_RED_G") ? atoi(getenv("YN_RED_G")) : 256;
    const dim3 grid(hreduce_blocks(a.M, 256 / a.lanes, gmax));
    if (mode == 0) hipLaunchKernelGGL(hcol_reduce_kernel<0>, grid, dim3(256), 0, s, a);
    else if (mode == 2) hipLaunchKernelGGL(hcol_reduce_kernel<2>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(hcol_reduce_kernel<3>, grid, dim3(256), 0, s, a);
}

// one row-octet of the BatchNorm forward output: dense (out has y's map, pads zero) or shuffle (out = the gapped unit output:
// out[2c] = pass[c], out[2c+1] = z[c], pads zeroed) — hbn_apply_kernel's store
__device__ __forceinline__ void bn_apply_emit(h16* __restrict__ out, int out_ld, int out_off, bool shuffle, int out_half, int out_gap, int C, float negslope, long m, int p0, int ol,
                                              const f32x2 (&mu)[4], const f32x2 (&is)[4], const f32x2 (&ga)[4], const f32x2 (&be)[4],       // pad channels: gamma = beta = 0
                                              const h16x8& v, const h16x8& pv)
{
    float z[8];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2 zz = bn_value2(bn_xhat2(pair_of(v, p), mu[p], is[p]), ga[p], be[p]);
        f32x2 mlt;
        mlt.x = zz.x > 0.0f ? 1.0f : negslope;
        mlt.y = zz.y > 0.0f ? 1.0f : negslope;
        const f32x2 r = zz * mlt + splat2(0.0f);                     // + 0: a ReLU'd negative is stored as +0, as max(z, 0) would
        z[2 * p] = r.x; z[2 * p + 1] = r.y;
    }
    if (!shuffle) {
        h16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (h16)z[j];
        sth8(out + (size_t)m * out_ld + out_off + p0, r);
    } else {
        h16* o = out + (size_t)m * out_ld;
        const int l0 = 2 * p0;                                      // logical position of the octet's first output pair (y dense: physical == logical)
        if (p0 + 8 <= C && (l0 + 16 <= out_half || l0 >= out_half)) {      // the 16 interleaved values do not straddle the plane boundary: two 16-byte stores
            const int pp = l0 + (l0 >= out_half ? out_gap : 0);
            h16x8 r0, r1;
#pragma unroll
            for (int j = 0; j < 4; ++j) { r0[2 * j] = pv[j]; r0[2 * j + 1] = (h16)z[j]; r1[2 * j] = pv[4 + j]; r1[2 * j + 1] = (h16)z[4 + j]; }
            sth8(o + pp, r0); sth8(o + pp + 8, r1);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = p0 + j;
                if (c >= C) continue;
                const int l = 2 * c;
                const int pp = l + (l >= out_half ? out_gap : 0);
                h16x2 w2; w2[0] = pv[j]; w2[1] = (h16)z[j];
                *reinterpret_cast<h16x2*>(o + pp) = w2;
            }
        }
        if (ol == 0 && out_gap > 0) {                               // the two pad runs of the gapped row
            for (int q = 0; q < out_gap; ++q) { o[out_half + q] = (h16)0.0f; o[2 * out_half + out_gap + q] = (h16)0.0f; }
        }
    }
}

// ---- BatchNorm forward apply (batch statistics): z = act((y - mean) * invstd * gamma + beta), h16 in / out.
//      Dense mode: out has y's channel map (pads written as zero).  Shuffle mode (pass != null): y dense [M][*] with C = bf
//      channels, out = the gapped 2*bf-channel unit output: out[2c] = pass[c], out[2c+1] = z[c]  (concat + channel_shuffle,
//      backbone/shufflenetv2.py:14-28,72-74), pads zeroed.  Block 0 saves mean / invstd and updates the running statistics.
__global__ __launch_bounds__(256) void hbn_apply_kernel(HBnApplyArgs a)
{
    // per-channel constants once per BLOCK (thread c <-> logical channel c; C <= 256), not per thread: mean / invstd from the
    // double sums (16 loads + a double sqrt per channel), gamma, beta; block 0 also saves them and updates the running statistics
    __shared__ float cst[4][256];
    const double invM = 1.0 / (double)a.M;
    for (int c = threadIdx.x; c < a.C; c += 256) {
        double m = 0.0, q = 0.0;
#pragma unroll
        for (int sl = 0; sl < HACC_SLOTS; ++sl) { m += a.acc[((size_t)sl * 2) * a.C + c]; q += a.acc[((size_t)sl * 2 + 1) * a.C + c]; }
        m *= invM;
        double var = q * invM - m * m;
        if (var < 0.0) var = 0.0;
        const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)a.eps));
        cst[0][c] = mu; cst[1][c] = is; cst[2][c] = a.gamma[c]; cst[3][c] = a.beta[c];
        if (blockIdx.x == 0) {
            a.mean[c] = mu; a.invstd[c] = is;
            if (a.rmean) {
                const float unbiased = (float)(a.M > 1 ? var * ((double)a.M / (double)(a.M - 1)) : var);
                a.rmean[c] = (1.0f - a.momentum) * a.rmean[c] + a.momentum * mu;
                a.rvar[c] = (1.0f - a.momentum) * a.rvar[c] + a.momentum * unbiased;
            }
        }
    }
    __syncthreads();
    const int OL = a.lanes, rowsPer = 256 / OL;
    const int ol = threadIdx.x & (OL - 1), rl = threadIdx.x / OL;
    const int OC = a.Cp >> 3;
    if (ol >= OC) return;
    const int p0 = ol * 8;
    f32x2 mu[4], is[4], ga[4], be[4];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = logical_of(p0 + j, a.C, a.half, a.gap);
        const float m_ = c >= 0 ? cst[0][c] : 0.0f, i_ = c >= 0 ? cst[1][c] : 0.0f, g_ = c >= 0 ? cst[2][c] : 0.0f, b_ = c >= 0 ? cst[3][c] : 0.0f;
        if (j & 1) { mu[j >> 1].y = m_; is[j >> 1].y = i_; ga[j >> 1].y = g_; be[j >> 1].y = b_; }
        else { mu[j >> 1].x = m_; is[j >> 1].x = i_; ga[j >> 1].x = g_; be[j >> 1].x = b_; }
    }
    const float negslope = act_negslope(a.act);
    auto emit = [&](long m, h16x8 v, h16x8 pv) {
        bn_apply_emit(a.out, a.out_ld, a.out_off, a.pass != nullptr, a.out_half, a.out_gap, a.C, negslope, m, p0, ol, mu, is, ga, be, v, pv);
    };
    constexpr int U = 4;
    const long step = (long)gridDim.x * rowsPer;
    long r = (long)blockIdx.x * rowsPer + rl;
    for (; r + (U - 1) * step < a.M; r += U * step) {
        h16x8 v[U], pv[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = ldh8(a.y + (size_t)(r + u * step) * a.y_ld + p0);
#pragma unroll
        for (int u = 0; u < U; ++u) pv[u] = a.pass ? ldh8(a.pass + (size_t)(r + u * step) * a.pass_ld + a.pass_off + p0) : zero8();
#pragma unroll
        for (int u = 0; u < U; ++u) emit(r + u * step, v[u], pv[u]);
    }
    for (; r < a.M; r += step) emit(r, ldh8(a.y + (size_t)r * a.y_ld + p0), a.pass ? ldh8(a.pass + (size_t)r * a.pass_ld + a.pass_off + p0) : zero8());
}

static int hstream_blocks(long M, int rowsPer)
{
    long b = (M + (long)rowsPer * 4 - 1) / ((long)rowsPer * 4);
    // two workgroups per CU: every workgroup of these launches first adds up the HACC_SLOTS double copies of its layer's sums (30 KB for 116
    // channels at 16 copies), so the prologue traffic grows with the grid - with 32 copies 7.79 ms per 608 / bs-32 step at 1 536 workgroups, 7.68 at
    // 768 and at 512; with 16 copies 7.42 / 7.37 / 7.29 / 7.25 / 7.25 / 7.26 ms at 1 536 / 1 024 / 768 / 512 / 384 / 256
    static const int cap = getenv("YN_STREAM_CAP") ? atoi(getenv("YN_STREAM_CAP")) : 256 * 2;
    if (b > cap) b = cap;
    if (b < 1) b = 1;
    return (int)b;
}

void launch_hbn_apply(const HBnApplyArgs& a0, hipStream_t s)
{
    HBnApplyArgs a = a0;
    a.lanes = hlanes_for(a.Cp);
    hipLaunchKernelGGL(hbn_apply_kernel, dim3((unsigned)hstream_blocks(a.M, 256 / a.lanes)), dim3(256), 0, s, a);
}

// this layer's gradient octet (channels p0..p0+7) of row `row`, as stored halves; `ev` (dz_odd only) gets the EVEN logical channels of
// the same 16 values — the pass-through half of the unit gradient (backbone/shufflenetv2.py:70-74 backwards)
__device__ __forceinline__ h16x8 dz_load_both(const HRedArgs& a, const DzOdd& dm, size_t row, int p0, h16x8& ev)
{
    if (!a.dz_odd) { ev = zero8(); return ldh8(a.dz + row * a.dz_ld + a.dz_off + p0); }
    h16x8 g;
    if (dm.vec) {
        const h16* q = a.dz + row * a.dz_ld;
        const h16x8 d0 = ldh8(q + dm.o0), d1 = ldh8(q + dm.o1), e = ldh8(q + dm.oe);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 2 * j + 1, k = 2 * j;
            g[j] = (dm.sv == 0 && i >= dm.cut) ? e[i] : d0[i];
            ev[j] = p0 + j < a.C ? ((dm.sv == 0 && k >= dm.cut) ? e[k] : d0[k]) : (h16)0.0f;
            g[4 + j] = dm.load1 ? ((dm.sv == 1 && i >= dm.cut) ? e[i] : d1[i]) : (h16)0.0f;
            ev[4 + j] = (dm.load1 && p0 + 4 + j < a.C) ? ((dm.sv == 1 && k >= dm.cut) ? e[k] : d1[k]) : (h16)0.0f;
        }
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = p0 + j;
            const int l = 2 * c + 1, k = 2 * c;
            g[j] = c < a.C ? a.dz[row * a.dz_ld + l + (l >= a.dz_half ? a.dz_gap : 0)] : (h16)0.0f;
            ev[j] = c < a.C ? a.dz[row * a.dz_ld + k + (k >= a.dz_half ? a.dz_gap : 0)] : (h16)0.0f;
        }
    }
    return g;
}

// ---- BatchNorm backward: dy = gamma * invstd * (dyh - mean(dyh) - xhat * mean(dyh * xhat)); dy dense h16 [M][Cp] with y's map,
//      pads zero; one thread per channel writes dgamma / dbeta (fp32, still carrying the loss scale).
__global__ __launch_bounds__(256) void hbn_bwd_kernel(HRedArgs a, h16* dy /* may be a.y: in place */, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    __shared__ float cst[6][256];                             // mean, invstd, gamma, beta, mean(dyh), mean(dyh * xhat) per logical channel
    const double invM = 1.0 / (double)a.M;
    for (int c = threadIdx.x; c < a.C; c += 256) {
        double s0 = 0.0, s1 = 0.0;
#pragma unroll
        for (int sl = 0; sl < HACC_SLOTS; ++sl) { s0 += a.acc[((size_t)sl * 2) * a.C + c]; s1 += a.acc[((size_t)sl * 2 + 1) * a.C + c]; }
        cst[0][c] = a.mean[c]; cst[1][c] = a.invstd[c]; cst[2][c] = a.gamma[c]; cst[3][c] = a.beta[c];
        cst[4][c] = (float)(s0 * invM); cst[5][c] = (float)(s1 * invM);
        if (blockIdx.x == 0) { dbeta[c] = (float)s0; dgamma[c] = (float)s1; }
    }
    __syncthreads();
    const int OL = a.lanes, rowsPer = 256 / OL;
    const int ol = threadIdx.x & (OL - 1), rl = threadIdx.x / OL;
    const int OC = a.Cp >> 3;
    if (ol >= OC) return;
    const int p0 = ol * 8;
    f32x2 mu[4], is[4], ga[4], be[4], m0[4], m1[4];              // pad channels: gamma = invstd = 0 -> dy = 0
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int c = logical_of(p0 + j, a.C, a.half, a.gap);
        const bool on = c >= 0;
        const float m_ = on ? cst[0][c] : 0.0f, i_ = on ? cst[1][c] : 0.0f, g_ = on ? cst[2][c] : 0.0f, b_ = on ? cst[3][c] : 0.0f;
        const float a_ = on ? cst[4][c] : 0.0f, c_ = on ? cst[5][c] : 0.0f;
        if (j & 1) { mu[j >> 1].y = m_; is[j >> 1].y = i_; ga[j >> 1].y = g_; be[j >> 1].y = b_; m0[j >> 1].y = a_; m1[j >> 1].y = c_; }
        else { mu[j >> 1].x = m_; is[j >> 1].x = i_; ga[j >> 1].x = g_; be[j >> 1].x = b_; m0[j >> 1].x = a_; m1[j >> 1].x = c_; }
    }
    const float negslope = act_negslope(a.act);
    DzOdd dm{};
    if (a.dz_odd) dm = dz_odd_map(a, p0);
    auto emit = [&](long m, const h16x8& v, const h16x8& g, const h16x8& ev) {
        h16x8 r;
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const f32x2 xh = bn_xhat2(pair_of(v, p), mu[p], is[p]);
            const f32x2 d = act_grad2(pair_of(g, p), bn_value2(xh, ga[p], be[p]), negslope);
            const f32x2 o = (ga[p] * is[p]) * (d - m0[p] - xh * m1[p]);
            r[2 * p] = (h16)o.x; r[2 * p + 1] = (h16)o.y;
        }
        sth8(dy + (size_t)m * a.Cp + p0, r);
        if (a.even) sth8(a.even + (size_t)m * a.even_ld + p0, ev);
    };
    const long step = (long)gridDim.x * rowsPer;
    long r = (long)blockIdx.x * rowsPer + rl;
    if (a.dz_odd && !dm.vec) {                                  // a ragged last octet whose 16-byte loads would leave the row (C = 5..7 mod 8: no layer of the networks)
        for (; r < a.M; r += step) {
            h16x8 ev;
            const h16x8 g = dz_load_both(a, dm, (size_t)r, p0, ev);
            emit(r, ldh8(a.y + (size_t)r * a.y_ld + a.y_off + p0), g, ev);
        }
        return;
    }
    constexpr int U = 4;
    const size_t dzo = a.dz_odd ? (size_t)dm.o0 : (size_t)(a.dz_off + p0);
    auto pick = [&](const h16x8& d0, const h16x8& d1, const h16x8& e, h16x8& g, h16x8& ev) {
        if (!a.dz_odd) { g = d0; ev = zero8(); return; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int i = 2 * j + 1, k = 2 * j;
            g[j] = (dm.sv == 0 && i >= dm.cut) ? e[i] : d0[i];
            ev[j] = p0 + j < a.C ? ((dm.sv == 0 && k >= dm.cut) ? e[k] : d0[k]) : (h16)0.0f;
            g[4 + j] = dm.load1 ? ((dm.sv == 1 && i >= dm.cut) ? e[i] : d1[i]) : (h16)0.0f;
            ev[4 + j] = (dm.load1 && p0 + 4 + j < a.C) ? ((dm.sv == 1 && k >= dm.cut) ? e[k] : d1[k]) : (h16)0.0f;
        }
    };
    for (; r + (U - 1) * step < a.M; r += U * step) {
        h16x8 v[U], d0[U], d1[U], e[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const size_t row = (size_t)(r + u * step);
            v[u] = ldh8(a.y + row * a.y_ld + a.y_off + p0);
            d0[u] = ldh8(a.dz + row * a.dz_ld + dzo);
            if (a.dz_odd) { d1[u] = ldh8(a.dz + row * a.dz_ld + dm.o1); e[u] = ldh8(a.dz + row * a.dz_ld + dm.oe); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            h16x8 g, ev;
            pick(d0[u], d1[u], e[u], g, ev);
            emit(r + u * step, v[u], g, ev);
        }
    }
    for (; r < a.M; r += step) {
        const size_t row = (size_t)r;
        const h16x8 v = ldh8(a.y + row * a.y_ld + a.y_off + p0), d0 = ldh8(a.dz + row * a.dz_ld + dzo);
        h16x8 d1 = d0, e = d0, g, ev;
        if (a.dz_odd) { d1 = ldh8(a.dz + row * a.dz_ld + dm.o1); e = ldh8(a.dz + row * a.dz_ld + dm.oe); }
        pick(d0, d1, e, g, ev);
        emit(r, v, g, ev);
    }
}

void launch_hbn_bwd(const HRedArgs& a0, h16* dy, float* dgamma, float* dbeta, hipStream_t s, bool sums_done)
{
    if (!sums_done) launch_hcol_reduce(a0, 2, s);
    HRedArgs a = a0;
    a.lanes = hlanes_for(a.Cp);
    hipLaunchKernelGGL(hbn_bwd_kernel, dim3((unsigned)hstream_blocks(a.M, 256 / a.lanes)), dim3(256), 0, s, a, dy, dgamma, dbeta);
}

// =================================================================================================
// Depthwise weight gradient: dW[c][tap] += sum_p dY[p][c] * X[p*stride + tap - 1][c]  (reference layout [C][1][3][3], logical channels).
// Block = OL octet-lanes x row-lanes; a row-lane takes a contiguous range of output pixels; the block's partial [C][9] goes to its own row of
// a scratch matrix, a second small kernel adds the rows into dW.
// =================================================================================================
template <int STRIDE>
__global__ __launch_bounds__(256) void hdw_wgrad_kernel(const h16* __restrict__ dy, int dy_ld, const h16* __restrict__ x, int x_ld, int x_off,
                                                         int B, int H, int W, int C, int Cp, int half, int gap, float* __restrict__ part, int OL)
{
    __shared__ float red[128][9];
    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const int rowsPer = 256 / OL;
    const int ol = threadIdx.x & (OL - 1), rl = threadIdx.x / OL;
    const int OC = Cp >> 3;
    const bool live = ol < OC;
    const int c0 = ol * 8;
    float acc[9][8];
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[k][j] = 0.0f;
    if (live) {
        // a row-lane walks RUNS of R output pixels along an image row: the 3 x NCOL input window of a run is loaded once (18 / 27
        // 16-byte loads for stride 1 / 2 instead of 9 per pixel), all loads of a run are issued before any FMA
        constexpr int R = 4, NCOL = STRIDE == 1 ? R + 2 : 2 * R + 1;
        const int runsPerRow = (Wo + R - 1) / R;
        const long nruns = (long)B * Ho * runsPerRow;
        const long lanes_total = (long)gridDim.x * rowsPer;
        const long per = (nruns + lanes_total - 1) / lanes_total;
        const long begin = ((long)blockIdx.x * rowsPer + rl) * per;
        const long end = begin + per < nruns ? begin + per : nruns;
        for (long u = begin; u < end; ++u) {
            const int seg = (int)(u % runsPerRow); const long row = u / runsPerRow;
            const int oy = (int)(row % Ho), b = (int)(row / Ho);
            const int ox0 = seg * R;
            h16x8 col[3][NCOL], g[R];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * STRIDE - 1 + ky;
                const bool yok = iy >= 0 && iy < H;
                const h16* xr = x + ((size_t)(b * H + (yok ? iy : 0)) * W) * x_ld + x_off + c0;
#pragma unroll
                for (int j = 0; j < NCOL; ++j) {
                    const int ix = ox0 * STRIDE - 1 + j;
                    col[ky][j] = keep8(ldh8(xr + (size_t)(ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * x_ld), yok && ix >= 0 && ix < W);
                }
            }
#pragma unroll
            for (int o = 0; o < R; ++o) {
                const int ox = ox0 + o;
                g[o] = keep8(ldh8(dy + ((size_t)row * Wo + (ox < Wo ? ox : Wo - 1)) * dy_ld + c0), ox < Wo);
            }
#pragma unroll
            for (int o = 0; o < R; ++o)
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[ky * 3 + kx][j] += (float)g[o][j] * (float)col[ky][o * STRIDE + kx][j];
        }
    }
    // combine: the row-lanes of a wave by xor-shuffles (same octet: OL lanes apart), the four waves through LDS one channel of the octet at a
    // time, and the block's [C][9] partial goes to ITS OWN row of `part` with plain stores (hdw_wgrad_sum_kernel adds the rows up).  The first
    // form sent every block's 9*C values to 8 gradient slots with fp32 atomics: the same-address chains capped the grid at a few hundred
    // blocks (24-368: most of the chip idle, 32-110 us per layer) and made more blocks SLOWER (2048 blocks: 3x).
    for (int off = 32; off >= OL; off >>= 1) {
#pragma unroll
        for (int k = 0; k < 9; ++k)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[k][j] += __shfl_xor(acc[k][j], off);
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* prow = part + (size_t)blockIdx.x * C * 9;
    for (int j = 0; j < 8; ++j) {
        __syncthreads();
        if (lane < OL) {
#pragma unroll
            for (int k = 0; k < 9; ++k) red[wave * 32 + lane][k] = acc[k][j];
        }
        __syncthreads();
        if (threadIdx.x < OL && live) {
            const int lc = logical_of(c0 + j, C, half, gap);
            if (lc >= 0) {
#pragma unroll
                for (int k = 0; k < 9; ++k) prow[(size_t)lc * 9 + k] = (red[ol][k] + red[32 + ol][k]) + (red[64 + ol][k] + red[96 + ol][k]);
            }
        }
    }
}

// dw[i] += sum over the G block rows of part[g][i]  (i < n = 9 * C); grid (ceil(n / 256), GY): block row y sums its slice of g and adds it with
// one atomic per element (GY same-address atomics: nothing)
__global__ __launch_bounds__(256) void hdw_wgrad_sum_kernel(const float* __restrict__ part, int G, int n, float* __restrict__ dw)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int per = (G + gridDim.y - 1) / gridDim.y;
    const int g0 = blockIdx.y * per, g1 = g0 + per < G ? g0 + per : G;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int g = g0;
    for (; g + 3 < g1; g += 4) {
        s0 += part[(size_t)g * n + i]; s1 += part[(size_t)(g + 1) * n + i]; s2 += part[(size_t)(g + 2) * n + i]; s3 += part[(size_t)(g + 3) * n + i];
    }
    for (; g < g1; ++g) s0 += part[(size_t)g * n + i];
    if (g1 > g0) atomicAdd(dw + i, (s0 + s1) + (s2 + s3));
}

void launch_hdw_wgrad(const h16* dy, int dy_ld, const h16* x, int x_ld, int x_off, int B, int H, int W, int C, int Cp, int half, int gap, int stride,
                      float* dw, float* part, size_t part_cap, hipStream_t s)
{
    const int OL = hlanes_for(Cp);
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long npix = (long)B * Ho * ((Wo + 3) / 4);                    // runs of 4 output pixels
    static const int runs = getenv("YN_DWW_RUNS") ? atoi(getenv("YN_DWW_RUNS")) : 4;
    long G = (npix + (256 / OL) * runs - 1) / ((256 / OL) * runs);
    static const int gmax = getenv("YN_DWW_G") ? atoi(getenv("YN_DWW_G")) : 2048;
    if (G > gmax) G = gmax;
    if ((size_t)G * C * 9 > part_cap) G = (long)(part_cap / ((size_t)C * 9));
    if (G < 1) G = 1;
    if (stride == 1) hipLaunchKernelGGL(hdw_wgrad_kernel<1>, dim3((unsigned)G), dim3(256), 0, s, dy, dy_ld, x, x_ld, x_off, B, H, W, C, Cp, half, gap, part, OL);
    else hipLaunchKernelGGL(hdw_wgrad_kernel<2>, dim3((unsigned)G), dim3(256), 0, s, dy, dy_ld, x, x_ld, x_off, B, H, W, C, Cp, half, gap, part, OL);
    const int n = C * 9;
    hipLaunchKernelGGL(hdw_wgrad_sum_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(G >= 64 ? 16 : 1)), dim3(256), 0, s, part, (int)G, n, dw);
}

// =================================================================================================
// Stem (backbone/shufflenetv2.py:109): 3x3 stride-2 conv 3 -> 24 reading the fp32 NCHW input, h16 NHWC output [M][24];
// thread = one output pixel x 8 output channels.  Its weight gradient: thread = (patch element r < 27, 8 output channels),
// a block walks a range of output pixels, LDS combine, fp32 atomics into the gradient slots.
// =================================================================================================
__global__ __launch_bounds__(256) void hstem_kernel(const float* __restrict__ x, int B, int H, int W, const float* __restrict__ w /*[27][24]*/,
                                                     const float* __restrict__ bias, h16* __restrict__ y)
{
    // thread = one output pixel x all 24 channels: its 27 input values in one batch of clamped, masked loads; the weights are
    // wave-uniform and arrive through scalar loads as SGPR operands of the FMAs (the first version — thread = pixel x 8 channels —
    // re-read the 27 inputs three times and fetched 216 weights per thread through the vector cache: 428 us at 608 x 608, bs 32).
    // The fma chain per channel (ci, ky, kx ascending) is unchanged.
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * Wo;
    const long p = (long)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (p >= total) return;
    const int ox = (int)(p % Wo); const long q = p / Wo;
    const int oy = (int)(q % Ho), b = (int)(q / Ho);
    float in[27];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy - 1 + ky;
            const bool yok = iy >= 0 && iy < H;
            const float* xr = x + (((size_t)b * 3 + ci) * H + (iy < 0 ? 0 : (iy >= H ? H - 1 : iy))) * W;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = 2 * ox - 1 + kx;
                unsigned mk = (yok && ix >= 0 && ix < W) ? 0xffffffffu : 0u;
                asm volatile("" : "+v"(mk));
                in[ci * 9 + ky * 3 + kx] = __uint_as_float(__float_as_uint(xr[ix < 0 ? 0 : (ix >= W ? W - 1 : ix)]) & mk);
            }
        }
    float acc[24];
#pragma unroll
    for (int j = 0; j < 24; ++j) acc[j] = bias ? bias[j] : 0.0f;
#pragma unroll
    for (int i = 0; i < 27; ++i)
#pragma unroll
        for (int j = 0; j < 24; ++j) acc[j] = __builtin_fmaf(in[i], w[i * 24 + j], acc[j]);
#pragma unroll
    for (int o = 0; o < 3; ++o) {
        h16x8 r;
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = (h16)acc[o * 8 + j];
        sth8(y + (size_t)p * 24 + o * 8, r);
    }
}

void launch_hstem(const float* x, int B, int H, int W, const float* w, const float* bias, h16* y, hipStream_t s)
{
    const long total = (long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
    hipLaunchKernelGGL(hstem_kernel, dim3(xcd_grid((unsigned)((total + 255) / 256))), dim3(256), 0, s, x, B, H, W, w, bias, y);
}

// dw[oc][r] = sum over output pixels p of patch[p][r] * dy[p][oc]  (r = (ci, ky, kx) < 27).  A block walks its range of pixels in chunks
// of 64: the chunk's dy rows (h16 -> fp32) and its 64 x 27 input patch go to LDS once (the next chunk's values are requested into
// registers before the current one is used), then thread = (pixel lane pg < 3, patch element r, channel octet og < 3) accumulates 8
// channels over the pixels pg, pg + 3, ... from LDS: one x value and 8 dy values per 8 FMAs.  (First version: the same thread roles
// reading straight from global — 81 threads x 2 loads per pixel, 27-fold redundant: 725 us at 608 x 608, bs 32, at the very end of
// the backward pass where nothing overlaps it.)
// MFMA = true (round 4): the accumulation loop - 66 LDS reads and 176 FMAs per thread and 64-pixel chunk, ~70 us of the kernel's 170 at the very end
// of the step - as dW[24 -> 32][27 -> 32] = dy^T x patch on the f16 MFMA: wavefront w takes the 16 pixels of k-step w of the chunk, gathers its
// dy fragment (exact: dy IS fp16) and its patch fragment from the same LDS tiles, splits the fp32 patch values into hi + lo * 2^-11 (two MFMAs,
// 2^-22 relative against the fp32 product) and keeps one 32 x 32 accumulator pair; the four wavefronts' tiles are added through LDS at the end.
template <bool MFMA>
__global__ __launch_bounds__(256) void hstem_wgrad_kernel(const h16* __restrict__ dy, const float* __restrict__ x, int B, int H, int W,
                                                           float* __restrict__ dw /* slots, reference layout [24][3][3][3] */, size_t slot_stride)
{
    constexpr int P = 64, NX = 7;                          // 4 threads x 7 patch elements per chunk pixel
    __shared__ __attribute__((aligned(16))) float xs[P * 27];
    __shared__ __attribute__((aligned(16))) float gs[P * 24];
    __shared__ __attribute__((aligned(16))) float red[256][8];
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int t = threadIdx.x;
    const long npix = (long)B * Ho * Wo;
    const long per = ((npix + gridDim.x - 1) / gridDim.x + P - 1) / P * P;
    const long begin = (long)blockIdx.x * per;
    const long end = begin + per < npix ? begin + per : npix;
    float xr[NX];
    h16x8 gr;
    // patch loads: thread = (chunk pixel t / 4, patch elements r = t % 4 + 4 k): one pixel decode per thread and chunk
    const int fpl = t >> 2, frq = t & 3;
    auto fetch = [&](long base) {
        const long pp = base + fpl;
        const int pc = (int)(pp < npix ? pp : npix - 1);
        const int ox = pc % Wo, q = pc / Wo;
        const int oy = q % Ho, b = q / Ho;
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int r = frq + 4 * i;
            const int ci = r / 9, k9 = r - ci * 9, ky = k9 / 3, kx = k9 - ky * 3;
            const int iy = 2 * oy - 1 + ky, ix = 2 * ox - 1 + kx;
            const bool ok = r < 27 && pp < end && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float v = x[(((size_t)b * 3 + (ci < 3 ? ci : 2)) * H + (iy < 0 ? 0 : (iy >= H ? H - 1 : iy))) * W + (ix < 0 ? 0 : (ix >= W ? W - 1 : ix))];
            unsigned mk = ok ? 0xffffffffu : 0u;
            asm volatile("" : "+v"(mk));
            xr[i] = __uint_as_float(__float_as_uint(v) & mk);
        }
        {
            const int pl = t / 3, o = t - pl * 3;
            const long pp = base + pl;
            const bool ok = t < P * 3 && pp < end;
            gr = ldh8(dy + (size_t)(ok ? pp : 0) * 24 + o * 8);
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) gr[j] = (h16)0.0f;
            }
        }
    };
    auto stage = [&]() {
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int r = frq + 4 * i;
            if (r < 27) xs[fpl * 27 + r] = xr[i];
        }
        if (t < P * 3) {
            *reinterpret_cast<float4*>(gs + t * 8) = make_float4((float)gr[0], (float)gr[1], (float)gr[2], (float)gr[3]);
            *reinterpret_cast<float4*>(gs + t * 8 + 4) = make_float4((float)gr[4], (float)gr[5], (float)gr[6], (float)gr[7]);
        }
    };
    const int pg = t / 81, u = t - pg * 81, r = u / 3, og = u - r * 3;
    const bool live = pg < 3;
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
    const int lane = t & 63, wave = t >> 6, l31 = lane & 31, hh = lane >> 5;
    f32x16 m0, m1;
#pragma unroll
    for (int k = 0; k < 16; ++k) { m0[k] = 0.0f; m1[k] = 0.0f; }
    if (begin < end) fetch(begin);
    for (long base = begin; base < end; base += P) {
        __syncthreads();                                    // the previous chunk has been consumed
        stage();
        __syncthreads();
        if (base + P < end) fetch(base + P);
        if (MFMA) {
            // k-step `wave` of the chunk: pixels p0 .. p0 + 7 of this half-wave; lane = (dy channel | patch element) l31
            const int p0 = wave * 16 + hh * 8;
            h16x8 av, bh, bl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float gv = l31 < 24 ? gs[(p0 + j) * 24 + l31] : 0.0f;
                const float xv = l31 < 27 ? xs[(p0 + j) * 27 + l31] : 0.0f;
                av[j] = (h16)gv;
                const h16 hi = (h16)xv;
                bh[j] = hi; bl[j] = (h16)((xv - (float)hi) * 2048.0f);
            }
            m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bh, m0, 0, 0, 0);
            m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bl, m1, 0, 0, 0);
        } else if (live) {
#pragma unroll 2
            for (int p = pg; p < P; p += 3) {
                const float xv = xs[p * 27 + r];
                const float4 g0 = *reinterpret_cast<const float4*>(gs + p * 24 + og * 8), g1 = *reinterpret_cast<const float4*>(gs + p * 24 + og * 8 + 4);
                acc[0] = __builtin_fmaf(xv, g0.x, acc[0]); acc[1] = __builtin_fmaf(xv, g0.y, acc[1]);
                acc[2] = __builtin_fmaf(xv, g0.z, acc[2]); acc[3] = __builtin_fmaf(xv, g0.w, acc[3]);
                acc[4] = __builtin_fmaf(xv, g1.x, acc[4]); acc[5] = __builtin_fmaf(xv, g1.y, acc[5]);
                acc[6] = __builtin_fmaf(xv, g1.z, acc[6]); acc[7] = __builtin_fmaf(xv, g1.w, acc[7]);
            }
        }
    }
    if (MFMA) {
        // m[k]: dy channel n = (k & 3) + 8 (k >> 2) + 4 hh, patch element l31; the four wavefronts' tiles through `red` ([256][8] floats) in two halves
        float* rf = &red[0][0];
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 8; ++k) rf[(wave * 64 + lane) * 8 + k] = __builtin_fmaf(m1[half * 8 + k], 1.0f / 2048.0f, m0[half * 8 + k]);
            __syncthreads();
            if (wave == 0 && l31 < 27 && begin < end) {
                float* out = dw + (size_t)(blockIdx.x & (GRAD_SLOTS - 1)) * slot_stride;
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const int kk = half * 8 + k, n = (kk & 3) + 8 * (kk >> 2) + 4 * hh;
                    if (n < 24) atomicAdd(out + (size_t)n * 27 + l31, (rf[lane * 8 + k] + rf[(64 + lane) * 8 + k]) + (rf[(128 + lane) * 8 + k] + rf[(192 + lane) * 8 + k]));
                }
            }
        }
        return;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) red[t][j] = acc[j];
    __syncthreads();
    if (t < 81 && begin < end) {
        float* out = dw + (size_t)(blockIdx.x & (GRAD_SLOTS - 1)) * slot_stride;
#pragma unroll
        for (int j = 0; j < 8; ++j) atomicAdd(out + (size_t)(og * 8 + j) * 27 + r, red[t][j] + red[t + 81][j] + red[t + 162][j]);
    }
}

void launch_hstem_wgrad(const h16* dy, const float* x, int B, int H, int W, float* dw_slots, size_t slot_stride, hipStream_t s)
{
    const long npix = (long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
    long G = (npix + 255) / 256;
    static const int gmax = getenv("YN_STEM_G") ? atoi(getenv("YN_STEM_G")) : 2048;
    if (G > gmax) G = gmax;
    if (G < 1) G = 1;
    static const int mfma = getenv("YN_STEM_WG_MFMA") ? atoi(getenv("YN_STEM_WG_MFMA")) : 1;        // 0: the FMA loop (A/B runs)
    if (mfma) hipLaunchKernelGGL(hstem_wgrad_kernel<true>, dim3((unsigned)G), dim3(256), 0, s, dy, x, B, H, W, dw_slots, slot_stride);
    else hipLaunchKernelGGL(hstem_wgrad_kernel<false>, dim3((unsigned)G), dim3(256), 0, s, dy, x, B, H, W, dw_slots, slot_stride);
}

// ---- 3x3 stride-2 max pool with recorded arg-max (first maximum in scan order, as ATen) and its gather-form backward; C = 24
// The argmax is kept as the WINDOW POSITION (ky * 3 + kx, one byte per channel: 8 bytes per octet instead of 32 for pixel indices —
// the backward kernel reads four windows' worth per input pixel).  First maximum in window order wins, as F.max_pool2d's index does.
__global__ __launch_bounds__(256) void hmaxpool_idx_kernel(const h16* __restrict__ x, int B, int H, int W, int Cp, h16* __restrict__ y, uint8_t* __restrict__ idx)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, OC = Cp >> 3;
    const long total = (long)B * Ho * Wo * OC;
    const long i = (long)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (i >= total) return;
    const int oc = (int)(i % OC);
    const long p = i / OC;
    const int ox = (int)(p % Wo); const long q = p / Wo;
    const int oy = (int)(q % Ho), b = (int)(q / Ho);
    h16x8 v[9];
    bool ok[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * 2 - 1 + ky;
        const bool yok = iy >= 0 && iy < H;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int ix = ox * 2 - 1 + kx;
            ok[ky * 3 + kx] = yok && ix >= 0 && ix < W;
            v[ky * 3 + kx] = ldh8(x + ((size_t)(b * H + (yok ? iy : 0)) * W + (ix < 0 ? 0 : (ix >= W ? W - 1 : ix))) * Cp + oc * 8);
        }
    }
    float m[8]; int best[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; best[j] = -1; }
#pragma unroll
    for (int k = 0; k < 9; ++k)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool take = ok[k] && ((float)v[k][j] > m[j] || best[j] < 0);
            m[j] = take ? (float)v[k][j] : m[j];
            best[j] = take ? k : best[j];
        }
    h16x8 r;
    unsigned lo = 0, hi = 0;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        r[j] = (h16)m[j];
        if (j < 4) lo |= (unsigned)(best[j] & 0xff) << (8 * j); else hi |= (unsigned)(best[j] & 0xff) << (8 * (j - 4));
    }
    *reinterpret_cast<uint2*>(idx + (size_t)p * Cp + oc * 8) = make_uint2(lo, hi);
    sth8(y + (size_t)p * Cp + oc * 8, r);
}

__global__ __launch_bounds__(256) void hmaxpool_bwd_kernel(const h16* __restrict__ dy, const uint8_t* __restrict__ idx, int B, int H, int W, int Cp, h16* __restrict__ dx)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, OC = Cp >> 3;
    const long total = (long)B * H * W * OC;
    const long i = (long)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;
    if (i >= total) return;
    const int oc = (int)(i % OC);
    const long p = i / OC;
    const int ix = (int)(p % W); const long q = p / W;
    const int iy = (int)(q % H), b = (int)(q / H);
    int oy[2], ox[2]; bool vy[2], vx[2];
    oy[0] = (iy + 1) >> 1; vy[0] = oy[0] < Ho; oy[1] = oy[0] - 1; vy[1] = (iy & 1) && oy[1] >= 0;
    ox[0] = (ix + 1) >> 1; vx[0] = ox[0] < Wo; ox[1] = ox[0] - 1; vx[1] = (ix & 1) && ox[1] >= 0;
    h16x8 g[4];
    uint2 cd[4];
    unsigned me[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = u >> 1, f = u & 1;
        const size_t o = ((size_t)(b * Ho + (vy[e] ? oy[e] : 0)) * Wo + (vx[f] ? ox[f] : 0)) * Cp + oc * 8;
        g[u] = ldh8(dy + o);
        cd[u] = *reinterpret_cast<const uint2*>(idx + o);
        me[u] = (vy[e] && vx[f]) ? (unsigned)((iy - (2 * oy[e] - 1)) * 3 + (ix - (2 * ox[f] - 1))) : 0xffu;      // this pixel's position in that window (0xff: no such window)
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned c = ((j < 4 ? cd[u].x : cd[u].y) >> (8 * (j & 3))) & 0xffu;
            acc[j] += c == me[u] ? (float)g[u][j] : 0.0f;
        }
    h16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (h16)acc[j];
    sth8(dx + (size_t)p * Cp + oc * 8, r);
}

void launch_hmaxpool_idx(const h16* x, int B, int H, int W, int Cp, h16* y, uint8_t* idx, hipStream_t s)
{
    const long total = (long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * (Cp >> 3);
    hipLaunchKernelGGL(hmaxpool_idx_kernel, dim3(xcd_grid((unsigned)((total + 255) / 256))), dim3(256), 0, s, x, B, H, W, Cp, y, idx);
}
void launch_hmaxpool_bwd(const h16* dy, const uint8_t* idx, int B, int H, int W, int Cp, h16* dx, hipStream_t s)
{
    const long total = (long)B * H * W * (Cp >> 3);
    hipLaunchKernelGGL(hmaxpool_bwd_kernel, dim3(xcd_grid((unsigned)((total + 255) / 256))), dim3(256), 0, s, dy, idx, B, H, W, Cp, dx);
}

// =================================================================================================
// The stem's BatchNorm + activation + max pool without the 142 MB tensors between them (608 x 608, bs 32; round 4).
//   forward  (hstem_apply_pool_kernel): a1 = maxpool3x3s2(act(BN(y))) and the arg-max positions straight from the stem conv's output y:
//            the normalised full-resolution tensor a0 is never stored (hbn_apply wrote it, hmaxpool_idx read it back: 284 MB);
//   backward (hstem_bwd_kernel<0 / 1>): the gradient of a0 is the max pool's gather of a1's gradient - four candidate windows per pixel
//            (hmaxpool_bwd_kernel's body) - and is formed where it is consumed, rounded to fp16 as the stored tensor was: <0> takes the
//            BatchNorm-backward sums, <1> writes dy in place over y (hbn_bwd_kernel's arithmetic).  hmaxpool_bwd's 142 MB write and
//            the two reads of it are gone.
// Thread = one channel octet (of 3: C = Cp = 24) of a pixel; a workgroup is 85 pixels x 3 octets and walks pixel blocks, so a thread's
// channel constants stay in registers.
// =================================================================================================
struct HStemCst { f32x2 mu[4], is[4], ga[4], be[4]; };
__device__ __forceinline__ void hstem_cst_load(const float (*cst)[32], int oc, HStemCst& k)
{
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int c = oc * 8 + 2 * p;
        k.mu[p].x = cst[0][c]; k.mu[p].y = cst[0][c + 1]; k.is[p].x = cst[1][c]; k.is[p].y = cst[1][c + 1];
        k.ga[p].x = cst[2][c]; k.ga[p].y = cst[2][c + 1]; k.be[p].x = cst[3][c]; k.be[p].y = cst[3][c + 1];
    }
}
__global__ __launch_bounds__(256) void hstem_apply_pool_kernel(HBnApplyArgs a, int B, int H, int W, h16* __restrict__ out, uint8_t* __restrict__ idx)
{
    constexpr int C = 24, PB = 85;
    __shared__ float cst[4][32];
    const double invM = 1.0 / (double)a.M;
    if (threadIdx.x < C) {                                      // hbn_apply_kernel's prologue: mean / invstd from the double sums; block 0 saves them
        const int c = threadIdx.x;
        double m = 0.0, q = 0.0;
#pragma unroll
        for (int sl = 0; sl < HACC_SLOTS; ++sl) { m += a.acc[((size_t)sl * 2) * C + c]; q += a.acc[((size_t)sl * 2 + 1) * C + c]; }
        m *= invM;
        double var = q * invM - m * m;
        if (var < 0.0) var = 0.0;
        const float mu = (float)m, is = (float)(1.0 / sqrt(var + (double)a.eps));
        cst[0][c] = mu; cst[1][c] = is; cst[2][c] = a.gamma[c]; cst[3][c] = a.beta[c];
        if (blockIdx.x == 0) {
            a.mean[c] = mu; a.invstd[c] = is;
            if (a.rmean) {
                const float unbiased = (float)(a.M > 1 ? var * ((double)a.M / (double)(a.M - 1)) : var);
                a.rmean[c] = (1.0f - a.momentum) * a.rmean[c] + a.momentum * mu;
                a.rvar[c] = (1.0f - a.momentum) * a.rvar[c] + a.momentum * unbiased;
            }
        }
    }
    __syncthreads();
    const int oc = threadIdx.x % 3, pl = threadIdx.x / 3;
    if (pl >= PB) return;
    HStemCst k;
    hstem_cst_load(cst, oc, k);
    const float negslope = act_negslope(a.act);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long npix = (long)B * Ho * Wo;
    for (long p = (long)blockIdx.x * PB + pl; p < npix; p += (long)gridDim.x * PB) {
        const int ox = (int)(p % Wo); const long q = p / Wo;
        const int oy = (int)(q % Ho), b = (int)(q / Ho);
        h16x8 v[9];
        bool ok[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            const bool yok = iy >= 0 && iy < H;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                ok[ky * 3 + kx] = yok && ix >= 0 && ix < W;
                v[ky * 3 + kx] = ldh8(a.y + ((size_t)(b * H + (yok ? iy : 0)) * W + (ix < 0 ? 0 : (ix >= W ? W - 1 : ix))) * C + oc * 8);
            }
        }
        float m[8]; int best[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) { m[j] = -INFINITY; best[j] = -1; }
#pragma unroll
        for (int t9 = 0; t9 < 9; ++t9) {
#pragma unroll
            for (int pp = 0; pp < 4; ++pp) {                    // bn_apply_emit's value, rounded to fp16 as the stored a0 was
                const f32x2 zz = bn_value2(bn_xhat2(pair_of(v[t9], pp), k.mu[pp], k.is[pp]), k.ga[pp], k.be[pp]);
                f32x2 mlt;
                mlt.x = zz.x > 0.0f ? 1.0f : negslope;
                mlt.y = zz.y > 0.0f ? 1.0f : negslope;
                const f32x2 r = zz * mlt + splat2(0.0f);
                const float z2[2] = {(float)(h16)r.x, (float)(h16)r.y};
#pragma unroll
                for (int e = 0; e < 2; ++e) {
                    const int j = 2 * pp + e;
                    const bool take = ok[t9] && (z2[e] > m[j] || best[j] < 0);      // first maximum in window order, as hmaxpool_idx_kernel
                    m[j] = take ? z2[e] : m[j];
                    best[j] = take ? t9 : best[j];
                }
            }
        }
        h16x8 r;
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            r[j] = (h16)m[j];
            if (j < 4) lo |= (unsigned)(best[j] & 0xff) << (8 * j); else hi |= (unsigned)(best[j] & 0xff) << (8 * (j - 4));
        }
        *reinterpret_cast<uint2*>(idx + (size_t)p * C + oc * 8) = make_uint2(lo, hi);
        sth8(out + (size_t)p * C + oc * 8, r);
    }
}

void launch_hstem_apply_pool(const HBnApplyArgs& a, int B, int H, int W, h16* out, uint8_t* idx, hipStream_t s)
{
    const long npix = (long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
    long G = (npix + 84) / 85;
    static const int gmax = getenv("YN_STEMPOOL_G") ? atoi(getenv("YN_STEMPOOL_G")) : 2048;
    if (G > gmax) G = gmax;
    hipLaunchKernelGGL(hstem_apply_pool_kernel, dim3((unsigned)(G < 1 ? 1 : G)), dim3(256), 0, s, a, B, H, W, out, idx);
}

// the max pool's input gradient at pixel (b, iy, ix), channel octet oc: hmaxpool_bwd_kernel's gather (H x W = the pool INPUT), fp16-rounded
__device__ __forceinline__ h16x8 hpool_gather(const h16* __restrict__ dy, const uint8_t* __restrict__ idx, int b, int iy, int ix, int H, int W, int Ho, int Wo, int Cp, int oc)
{
    int oy[2], ox[2]; bool vy[2], vx[2];
    oy[0] = (iy + 1) >> 1; vy[0] = oy[0] < Ho; oy[1] = oy[0] - 1; vy[1] = (iy & 1) && oy[1] >= 0;
    ox[0] = (ix + 1) >> 1; vx[0] = ox[0] < Wo; ox[1] = ox[0] - 1; vx[1] = (ix & 1) && ox[1] >= 0;
    h16x8 g[4];
    uint2 cd[4];
    unsigned me[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int e = u >> 1, f = u & 1;
        const size_t o = ((size_t)(b * Ho + (vy[e] ? oy[e] : 0)) * Wo + (vx[f] ? ox[f] : 0)) * Cp + oc * 8;
        g[u] = ldh8(dy + o);
        cd[u] = *reinterpret_cast<const uint2*>(idx + o);
        me[u] = (vy[e] && vx[f]) ? (unsigned)((iy - (2 * oy[e] - 1)) * 3 + (ix - (2 * ox[f] - 1))) : 0xffu;
    }
    float acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[j] = 0.0f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned c = ((j < 4 ? cd[u].x : cd[u].y) >> (8 * (j & 3))) & 0xffu;
            acc[j] += c == me[u] ? (float)g[u][j] : 0.0f;
        }
    h16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (h16)acc[j];
    return r;
}

// a: y / mean / invstd / gamma / beta / act / acc (the BACKWARD sums) / M of the stem's BatchNorm; dz = gather(pool gradient g1, idx)
template <int PHASE>                                            // 0: the sums -> acc;  1: dy (may be y: in place), dgamma / dbeta
__global__ __launch_bounds__(256) void hstem_bwd_kernel(HRedArgs a, const h16* __restrict__ g1, const uint8_t* __restrict__ idx, int B, int H, int W,
                                                         h16* dy, float* __restrict__ dgamma, float* __restrict__ dbeta)
{
    constexpr int C = 24, PB = 85;
    __shared__ float cst[6][32];
    const double invM = 1.0 / (double)a.M;
    if (threadIdx.x < C) {
        const int c = threadIdx.x;
        cst[0][c] = a.mean[c]; cst[1][c] = a.invstd[c]; cst[2][c] = a.gamma[c]; cst[3][c] = a.beta[c];
        if (PHASE == 1) {
            double s0 = 0.0, s1 = 0.0;
#pragma unroll
            for (int sl = 0; sl < HACC_SLOTS; ++sl) { s0 += a.acc[((size_t)sl * 2) * C + c]; s1 += a.acc[((size_t)sl * 2 + 1) * C + c]; }
            cst[4][c] = (float)(s0 * invM); cst[5][c] = (float)(s1 * invM);
            if (blockIdx.x == 0) { dbeta[c] = (float)s0; dgamma[c] = (float)s1; }
        }
    }
    __syncthreads();
    const int oc = threadIdx.x % 3, pl = threadIdx.x / 3;
    const bool worker = pl < PB;
    HStemCst k;
    hstem_cst_load(cst, oc, k);
    f32x2 m0[4], m1[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        m0[p] = m1[p] = splat2(0.0f);
        if (PHASE == 1) { const int c = oc * 8 + 2 * p; m0[p].x = cst[4][c]; m0[p].y = cst[4][c + 1]; m1[p].x = cst[5][c]; m1[p].y = cst[5][c + 1]; }
    }
    const float negslope = act_negslope(a.act);
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long npix = (long)B * H * W;
    double s0[8], s1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s0[j] = 0.0; s1[j] = 0.0; }
    f32x2 t0[4], t1[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) { t0[p] = splat2(0.0f); t1[p] = splat2(0.0f); }
    int batch = 0;
    if (worker) {
        for (long pi = (long)blockIdx.x * PB + pl; pi < npix; pi += (long)gridDim.x * PB) {
            const int ix = (int)(pi % W); const long q = pi / W;
            const int iy = (int)(q % H), b = (int)(q / H);
            const h16x8 v = ldh8(a.y + (size_t)pi * C + oc * 8);
            const h16x8 g = hpool_gather(g1, idx, b, iy, ix, H, W, Ho, Wo, C, oc);
            h16x8 r;
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                const f32x2 xh = bn_xhat2(pair_of(v, p), k.mu[p], k.is[p]);
                const f32x2 d = act_grad2(pair_of(g, p), bn_value2(xh, k.ga[p], k.be[p]), negslope);
                if (PHASE == 0) { t0[p] += d; t1[p] = __builtin_elementwise_fma(d, xh, t1[p]); }
                else { const f32x2 o = (k.ga[p] * k.is[p]) * (d - m0[p] - xh * m1[p]); r[2 * p] = (h16)o.x; r[2 * p + 1] = (h16)o.y; }
            }
            if (PHASE == 1) sth8(dy + (size_t)pi * C + oc * 8, r);
            else if (++batch == 4) {                            // fp32 over four pixels, then into the double accumulators (hcol_reduce_kernel's batches)
                batch = 0;
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    s0[2 * p] += (double)t0[p].x; s0[2 * p + 1] += (double)t0[p].y; s1[2 * p] += (double)t1[p].x; s1[2 * p + 1] += (double)t1[p].y;
                    t0[p] = splat2(0.0f); t1[p] = splat2(0.0f);
                }
            }
        }
    }
    if (PHASE == 0) {
#pragma unroll
        for (int p = 0; p < 4; ++p) { s0[2 * p] += (double)t0[p].x; s0[2 * p + 1] += (double)t0[p].y; s1[2 * p] += (double)t1[p].x; s1[2 * p + 1] += (double)t1[p].y; }
        // the workgroup's 85 pixel lanes -> one double per (sum, channel): fp32 is not enough here (a lane has added ~70 values), so the
        // lanes are combined in double, through LDS as two floats (hi + lo)
        __shared__ double dred[PB * 3 * 2];
        double tot[16];
#pragma unroll
        for (int j = 0; j < 8; ++j) { tot[j] = s0[j]; tot[8 + j] = s1[j]; }
        for (int kk = 0; kk < 16; kk += 2) {                    // two of the 16 values per round: 4 KB of LDS
            __syncthreads();
            if (worker) { dred[(pl * 3 + oc) * 2] = tot[kk]; dred[(pl * 3 + oc) * 2 + 1] = tot[kk + 1]; }
            __syncthreads();
            if (threadIdx.x < 6) {                              // (octet, which of the two)
                const int o3 = threadIdx.x >> 1, w = threadIdx.x & 1;
                double s = 0.0;
                for (int r2 = 0; r2 < PB; ++r2) s += dred[(r2 * 3 + o3) * 2 + w];
                const int kidx = kk + w, c = o3 * 8 + (kidx & 7);
                double* acc = a.acc + (size_t)(blockIdx.x & (HACC_SLOTS - 1)) * 2 * C;
                atomicAdd(acc + (kidx >> 3) * C + c, s);
            }
        }
    }
}

void launch_hstem_bwd(const HRedArgs& a, const h16* g1, const uint8_t* idx, int B, int H, int W, h16* dy, float* dgamma, float* dbeta, hipStream_t s)
{
    const long npix = (long)B * H * W;
    static const int g0 = getenv("YN_STEMBWD_G0") ? atoi(getenv("YN_STEMBWD_G0")) : 2048, g1n = getenv("YN_STEMBWD_G1") ? atoi(getenv("YN_STEMBWD_G1")) : 2048;
    long G = (npix + 84) / 85;
    const long G0 = G > g0 ? g0 : G, G1 = G > g1n ? g1n : G;
    hipLaunchKernelGGL(hstem_bwd_kernel<0>, dim3((unsigned)G0), dim3(256), 0, s, a, g1, idx, B, H, W, dy, dgamma, dbeta);
    hipLaunchKernelGGL(hstem_bwd_kernel<1>, dim3((unsigned)G1), dim3(256), 0, s, a, g1, idx, B, H, W, dy, dgamma, dbeta);
}

// ---- element-wise glue of the FPN / PAN adds (models/yolo_nano.py:291-296), dense Cp-channel h16 tensors; modes as resample_kernel
__global__ __launch_bounds__(256) void hresample_kernel(const h16* __restrict__ a, const h16* __restrict__ b, h16* __restrict__ out, int B, int H, int W, int Cp, int mode)
{
    const int OC = Cp >> 3;
    if (mode <= 1 || mode == 3) {
        const long total = (long)B * H * W * OC;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int oc = (int)(i % OC);
            const long p = i / OC;
            const int x = (int)(p % W); const long q = p / W;
            const int y = (int)(q % H), bb = (int)(q / H);
            size_t j;
            if (mode == 0) j = (((size_t)bb * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1));
            else           j = (((size_t)bb * (H << 1) + (y << 1)) * (W << 1) + (x << 1));
            const h16x8 va = ldh8(a + (size_t)p * Cp + oc * 8);
            h16x8 r;
            if (mode == 3) {                                            // out = grad of the high-res tensor, a = g: out[even pixel] += g
                const h16x8 vo = ldh8(out + j * Cp + oc * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = (h16)((float)vo[k] + (float)va[k]);
                sth8(out + j * Cp + oc * 8, r);
            } else {
                const h16x8 vb = ldh8(b + j * Cp + oc * 8);
#pragma unroll
                for (int k = 0; k < 8; ++k) r[k] = (h16)((float)va[k] + (float)vb[k]);
                sth8(out + (size_t)p * Cp + oc * 8, r);
            }
        }
    } else {                                                            // mode 2: out [B,H/2,W/2] += the 4 children of a [B,H,W]
        const int h2 = H >> 1, w2 = W >> 1;
        const long total = (long)B * h2 * w2 * OC;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int oc = (int)(i % OC);
            const long p = i / OC;
            const int x = (int)(p % w2); const long q = p / w2;
            const int y = (int)(q % h2), bb = (int)(q / h2);
            const size_t base = (((size_t)bb * H + 2 * y) * W + 2 * x) * Cp + oc * 8;
            const h16x8 c0 = ldh8(a + base), c1 = ldh8(a + base + Cp), c2 = ldh8(a + base + (size_t)W * Cp), c3 = ldh8(a + base + (size_t)W * Cp + Cp);
            const h16x8 vo = ldh8(out + (size_t)p * Cp + oc * 8);
            h16x8 r;
#pragma unroll
            for (int k = 0; k < 8; ++k) r[k] = (h16)((float)vo[k] + (((float)c0[k] + (float)c1[k]) + ((float)c2[k] + (float)c3[k])));
            sth8(out + (size_t)p * Cp + oc * 8, r);
        }
    }
}

void launch_hresample(const h16* a, const h16* b, h16* out, int B, int H, int W, int Cp, int mode, hipStream_t s)
{
    long n = (long)B * H * W * (Cp >> 3);
    if (mode == 2) n /= 4;
    long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(hresample_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, out, B, H, W, Cp, mode);
}

// dst[m][dp(j)] = src[m][sp(j)] for j < n, where sp(j) = phys_src(src_off + j*src_cs), dp(j) = phys_dst(dst_off + j*dst_cs) (gapped maps);
// dst pads [n, npad) of a dense destination are zeroed.  Used for: the pass-through half of the unit gradient (even logical
// channels of the gapped output gradient -> first plane of the input gradient), branch1's gradient, plain copies.
__global__ __launch_bounds__(256) void hgather_kernel(const h16* __restrict__ src, int src_ld, int src_off, int src_cs, int src_half, int src_gap,
                                                       h16* __restrict__ dst, int dst_ld, int dst_off, int dst_cs, int dst_half, int dst_gap, long M, int n, int npad)
{
    const long total = M * npad;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = (int)(i % npad);
        const long m = i / npad;
        const int dl = dst_off + j * dst_cs;
        const int dp = dl + (dl >= dst_half ? dst_gap : 0);
        h16 v = (h16)0.0f;
        if (j < n) {
            const int sl = src_off + j * src_cs;
            v = src[(size_t)m * src_ld + sl + (sl >= src_half ? src_gap : 0)];
        }
        dst[(size_t)m * dst_ld + dp] = v;
    }
}

void launch_hgather(const h16* src, int src_ld, int src_off, int src_cs, int src_half, int src_gap,
                    h16* dst, int dst_ld, int dst_off, int dst_cs, int dst_half, int dst_gap, long M, int n, int npad, hipStream_t s)
{
    long blocks = (M * npad + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(hgather_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, src_ld, src_off, src_cs, src_half, src_gap,
                       dst, dst_ld, dst_off, dst_cs, dst_half, dst_gap, M, n, npad);
}

// =================================================================================================
// Weight packing from the fp32 master weights (reference layouts), every step.
//   GEMM forward  : Wp[tap][kp/8][n][kp%8] = W[n][ci][tap], kp = phys_in(ci)        (rows at pad positions stay zero)
//   GEMM backward : Wp[tap'][np/8][kp_out][np%8] with the roles swapped: contraction over the OUTPUT channels n (physical, dense
//                   map of dY), result columns = physical INPUT channels; dense 3x3 taps flipped (tap' = 8 - tap)
//   depthwise     : Wd[tap][cp] = W[c][tap] (forward) / Wd[8 - tap][cp] (stride-1 input gradient), cp = phys(c)
// =================================================================================================
__global__ __launch_bounds__(256) void hpack_gemm_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, int in_half, int in_gap,
                                                          int Kp, int Npad, int backward, h16* __restrict__ out)
{
    const long total = (long)Cout * Cin * taps;
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int n = (int)(i / ((long)Cin * taps));
    const int rem = (int)(i - (long)n * Cin * taps);
    const int ci = rem / taps, tap = rem - ci * taps;
    const int cp = ci + (ci >= in_half ? in_gap : 0);
    if (!backward) out[(((size_t)tap * (Kp >> 3) + (cp >> 3)) * Npad + n) * 8 + (cp & 7)] = (h16)w[i];
    else out[(((size_t)(taps - 1 - tap) * (Kp >> 3) + (n >> 3)) * Npad + cp) * 8 + (n & 7)] = (h16)w[i];     // Kp = physical Cout, Npad covers physical Cin
}

__global__ __launch_bounds__(256) void hpack_dw_kernel(const float* __restrict__ w, const float* __restrict__ bias, int C, int half, int gap, int Cp, int flip,
                                                        float* __restrict__ out, float* __restrict__ bias_out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < C * 9) {
        const int c = i / 9, tap = i - c * 9;
        out[(size_t)(flip ? 8 - tap : tap) * Cp + c + (c >= half ? gap : 0)] = w[i];
    }
    if (bias_out && i < C) bias_out[i + (i >= half ? gap : 0)] = bias ? bias[i] : 0.0f;
}

__global__ __launch_bounds__(256) void hpack_stem_kernel(const float* __restrict__ w /*[24][3][3][3]*/, float* __restrict__ out /*[27][24]*/)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 24 * 27) return;
    const int co = i / 27, r = i - co * 27;
    out[(size_t)r * 24 + co] = w[i];
}

// every layer's packs in ONE launch (blockIdx.y = layer): the per-layer kernels above cost ~5 us of launch latency each, 150 per step
__global__ __launch_bounds__(256) void hpack_all_kernel(const HPackDesc* __restrict__ table)
{
    const HPackDesc d = table[blockIdx.y];
    const float* __restrict__ w = d.w;
    if (d.kind == 0) {
        const long total = (long)d.Cout * d.Cin * d.taps;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int n = (int)(i / ((long)d.Cin * d.taps));
            const int rem = (int)(i - (long)n * d.Cin * d.taps);
            const int ci = rem / d.taps, tap = rem - ci * d.taps;
            const int cp = ci + (ci >= d.half ? d.gap : 0);
            const h16 v = (h16)w[i];
            d.wf[(((size_t)tap * (d.Kp >> 3) + (cp >> 3)) * d.Npad + n) * 8 + (cp & 7)] = v;
            d.wb[(((size_t)(d.taps - 1 - tap) * (d.Kpb >> 3) + (n >> 3)) * d.Npadb + cp) * 8 + (n & 7)] = v;
        }
        if (d.b) for (int i = blockIdx.x * 256 + threadIdx.x; i < d.Cout; i += gridDim.x * 256) d.bias[i] = d.b[i];
    } else if (d.kind == 1) {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < d.Cout * 9; i += gridDim.x * 256) {
            const int c = i / 9, tap = i - c * 9;
            const int cp = c + (c >= d.half ? d.gap : 0);
            d.dwf[(size_t)tap * d.Kp + cp] = w[i];
            if (d.dwb) d.dwb[(size_t)(8 - tap) * d.Kp + cp] = w[i];
        }
        for (int i = blockIdx.x * 256 + threadIdx.x; i < d.Cout; i += gridDim.x * 256) d.bias[i + (i >= d.half ? d.gap : 0)] = d.b ? d.b[i] : 0.0f;
    } else {
        for (int i = blockIdx.x * 256 + threadIdx.x; i < 24 * 27; i += gridDim.x * 256) { const int co = i / 27, r = i - co * 27; d.dwf[(size_t)r * 24 + co] = w[i]; }
    }
}
void launch_hpack_all(const HPackDesc* table_dev, int n, hipStream_t s)
{
    hipLaunchKernelGGL(hpack_all_kernel, dim3(16, (unsigned)n), dim3(256), 0, s, table_dev);
}

void launch_hpack_gemm(const float* w, int Cout, int Cin, int taps, int in_half, int in_gap, int Kp, int Npad, int backward, h16* out, hipStream_t s)
{
    const long total = (long)Cout * Cin * taps;
    hipLaunchKernelGGL(hpack_gemm_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, Cout, Cin, taps, in_half, in_gap, Kp, Npad, backward, out);
}
void launch_hpack_dw(const float* w, const float* bias, int C, int half, int gap, int Cp, int flip, float* out, float* bias_out, hipStream_t s)
{
    hipLaunchKernelGGL(hpack_dw_kernel, dim3((unsigned)((C * 9 + 255) / 256)), dim3(256), 0, s, w, bias, C, half, gap, Cp, flip, out, bias_out);
}
void launch_hpack_stem(const float* w, float* out, hipStream_t s)
{
    hipLaunchKernelGGL(hpack_stem_kernel, dim3(3), dim3(256), 0, s, w, out);
}

// fp32 dense rows [M][C] <-> h16 rows in the padded / gapped layout (op-level parity entry points; dst pads pre-zeroed by the caller)
__global__ __launch_bounds__(256) void hstage_kernel(const float* __restrict__ src, int C, h16* __restrict__ dst, int ld, int half, int gap, long M)
{
    const long total = M * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        dst[(size_t)m * ld + c + (c >= half ? gap : 0)] = (h16)src[i];
    }
}
__global__ __launch_bounds__(256) void hunstage_kernel(const h16* __restrict__ src, int ld, int half, int gap, float* __restrict__ dst, int C, long M)
{
    const long total = M * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / C;
        const int c = (int)(i - m * C);
        dst[i] = (float)src[(size_t)m * ld + c + (c >= half ? gap : 0)];
    }
}
void launch_hstage(const float* src, int C, h16* dst, int ld, int half, int gap, long M, hipStream_t s)
{
    long blocks = (M * C + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(hstage_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, C, dst, ld, half, gap, M);
}
void launch_hunstage(const h16* src, int ld, int half, int gap, float* dst, int C, long M, hipStream_t s)
{
    long blocks = (M * C + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(hunstage_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, ld, half, gap, dst, C, M);
}

// dst[m][j] = (float)src[m*src_ld + j], j < n: the raw heads of a train-mode forward as dense fp32 rows (yn_train_forward)
template <typename T>
__global__ __launch_bounds__(256) void rows_to_f32_kernel(const T* __restrict__ src, int src_ld, float* __restrict__ dst, int n, long M)
{
    const long total = M * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const long m = i / n;
        dst[i] = (float)src[(size_t)m * src_ld + (int)(i - m * n)];
    }
}
void launch_rows_to_f32(const void* src, int is_h16, int src_ld, float* dst, int n, long M, hipStream_t s)
{
    long blocks = (M * n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    if (is_h16) hipLaunchKernelGGL(rows_to_f32_kernel<h16>, dim3((unsigned)blocks), dim3(256), 0, s, (const h16*)src, src_ld, dst, n, M);
    else hipLaunchKernelGGL(rows_to_f32_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, s, (const float*)src, src_ld, dst, n, M);
}

// =================================================================================================
// Loss-scale bookkeeping, all on the device (no host round trip, capture-friendly).
//   state[0] = current scale S, state[1] = 1/S, state[2] = clean steps since the last change (as float), state[3] = overflow flag of
//   THIS rank's last backward pass, state[4] = 1 while that pass has not been accounted for yet.
// hgrad_finish_kernel: g = (g + sum of the atomics slots) / S, raises the overflow flag when a value is not finite and marks the step
// pending.  The scale itself moves in hscale_update_kernel: overflow -> S /= 2 (>= 1), counter = 0; else counter++, and S *= 2 (<= 65536)
// after 2000 clean steps.  WHO decides "overflow": yn_sgd_step, from the finite-scan of the bucket it is about to apply (global != null)
// - under data parallelism that is the ALL-REDUCED bucket, non-finite on every rank as soon as one rank overflowed, so every replica
// halves (or grows) its scale on the same step and the scales never drift apart; a caller that applies the gradients with another
// optimiser (torch.optim.SGD on the flat buffers) never reaches yn_sgd_step, and the next fp16 step then settles the pending one from
// the local flag.  The overflowed step's gradients stay non-finite in the bucket, so yn_sgd_step skips it.
// =================================================================================================
__global__ __launch_bounds__(256) void hgrad_finish_kernel(float* __restrict__ g, const float* __restrict__ slots, long n, size_t stride, float* __restrict__ state)
{
    const float inv = state[1];
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = g[i];
#pragma unroll
        for (int s = 0; s < GRAD_SLOTS; ++s) v += slots[(size_t)s * stride + i];
        v *= inv;
        bad |= !(fabsf(v) <= 3.0e38f);
        g[i] = v;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(reinterpret_cast<unsigned*>(state + 3), 1u);
    if (blockIdx.x == 0 && threadIdx.x == 0) state[4] = 1.0f;
}

// global: the bucket-wide non-finite flag of yn_sgd_step (int[2], [0]) or null = settle a still-pending step from the local flag
__global__ void hscale_update_kernel(float* __restrict__ state, const int* __restrict__ global)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (state[4] == 0.0f) return;                          // nothing pending (already settled, or no fp16 step since)
    float S = state[0], clean = state[2];
    const unsigned over = global ? (unsigned)global[0] : *reinterpret_cast<unsigned*>(state + 3);
    if (over) { S = S > 1.0f ? S * 0.5f : 1.0f; clean = 0.0f; }
    else { clean += 1.0f; if (clean >= 2000.0f) { S = S < 65536.0f ? S * 2.0f : S; clean = 0.0f; } }
    state[0] = S; state[1] = 1.0f / S; state[2] = clean;
    *reinterpret_cast<unsigned*>(state + 3) = 0u;
    state[4] = 0.0f;
}

void launch_hgrad_finish(float* g, const float* slots, long n, size_t stride, float* state, hipStream_t s)
{
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(hgrad_finish_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, slots, n, stride, state);
}

void launch_hscale_update(float* state, const int* global_flag, hipStream_t s)
{
    hipLaunchKernelGGL(hscale_update_kernel, dim3(1), dim3(64), 0, s, state, global_flag);
}

}  // namespace ynk

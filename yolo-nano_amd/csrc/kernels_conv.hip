// kernels_conv.hip — gfx950 convolution kernels of the YOLO-Nano hot path (float32, NHWC).
//
//   gemm_conv_kernel   pointwise 1x1 and dense 3x3 convolutions as GEMMs on the f32 MFMA
//                      (v_mfma_f32_32x32x2_f32), LDS double-buffered, fused bias + activation,
//                      fused FPN/PAN resample-add prologue (3x3), fused concat+channel-shuffle
//                      epilogue (pointwise).
//   dwconv3x3_kernel   depthwise 3x3 (stride 1/2), one thread per (pixel, channel pair)
//   stem_kernel        3->24 3x3 stride-2 conv reading NCHW input, writing NHWC
//   maxpool_kernel     3x3 stride-2 max pool
//   (+ gemm_direct_kernel, conv3x3_halo*_kernel, stem_pool_kernel: see their headers;
//    the multi-layer tile kernels live in kernels_chain.hip, the shared device helpers in yn_device.h)
//
// Reference semantics: backbone/shufflenetv2.py:31-78,109-116, utils/modules.py:8-18,
// models/yolo_nano.py:286-301 — with BatchNorm folded into the weights (utils/fuse_conv_bn.py:6-22).
#include "yn_internal.h"
#include "yn_device.h"

#include <cstdio>
#include <mutex>
#include <set>
#include <string>

namespace ynk {

static thread_local const char* g_last_kernel = "";
const char* last_kernel_name() { return g_last_kernel; }
void set_last_kernel_name(const char* n) { g_last_kernel = n; }

bool g_log_lds = getenv("YN_LOG_LDS") != nullptr && atoi(getenv("YN_LOG_LDS")) != 0;
void note_launch_lds(const char* kernel, size_t dyn_lds, unsigned threads)
{
    static std::mutex mu;
    static std::set<std::string> seen;
    char line[512];
    snprintf(line, sizeof line, "yn_lds %s %zu %u", kernel, dyn_lds, threads);
    std::lock_guard<std::mutex> lk(mu);
    if (seen.insert(line).second) fprintf(stderr, "%s\n", line);
}

// -------------------------------------------------------------------------------------------------
// GEMM convolution.  Block = 4 waves laid out WM x WN; each wave owns a 32 x (32*NT) output tile and
// keeps it in NT 32x32 f32 MFMA accumulators.  K is consumed in chunks of 2*KP through NBUF LDS buffers:
//   As[kp][row] float2  (k-pair major; +1 float2 pad per kp row => conflict-free ds_write_b64)
//   Bs[kp][n]   float2  (straight copy of the packed weights)
// One ds_read_b64 of A and of B per lane feeds two MFMAs: lanes 0-31 carry k = 4q, 4q+1 and lanes
// 32-63 carry k = 4q+2, 4q+3 (the order of the k-sum inside a chunk is free as long as A and B agree;
// it is the same for every tile configuration, so all configurations are bit-identical).
// NBUF = 2: one barrier per chunk, the next chunk's global loads fly during the MFMAs.
// NBUF = 1: half the LDS (more blocks per CU); K <= 2*KP needs no loop at all.
// -------------------------------------------------------------------------------------------------
template <int WM, int WN, int NT, int MODE, int KP, int NBUF>
__global__ __launch_bounds__(256) void gemm_conv_kernel(GemmArgs a)
{
    constexpr int BM = 32 * WM, BN = 32 * NT * WN;
    constexpr int AS = BM * 2 + 2;              // floats per kp row of A
    constexpr int BS = BN * 2;                  // floats per kp row of B
    constexpr int A_PER = BM * KP / 256;        // float2 per thread per chunk
    constexpr int B_PER = KP * BN / 512;        // float4 per thread per chunk
    constexpr int RPP = 256 / KP;               // A rows per pass
    constexpr int NQ = KP / 2;                  // MFMA k-steps (of 4 k) per chunk
    static_assert(A_PER >= 1 && B_PER >= 1, "tile too small for 256 threads");
    __shared__ __attribute__((aligned(16))) float smem[NBUF * KP * (AS + BS)];
    float* As = smem;
    float* Bs = smem + NBUF * KP * AS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    // 1-D grid of (row tiles rounded up to a multiple of 8) x (column tiles), decoded XCD-aware (workgroup b runs on XCD
    // b % 8): every XCD owns a contiguous eighth of the row tiles and runs ALL column tiles of a row tile back to back, so
    // the A rows are fetched into one L2 once (with a 2-D grid the column tiles of a row land on different XCDs and A is
    // fetched once per column tile: 2.2x the algorithmic bytes on the stage-3 layers in the PMC pass).
    const unsigned gy = (unsigned)(a.Npad + BN - 1) / BN, gx8 = gridDim.x / gy;
    const unsigned slot = blockIdx.x >> 3;
    const int m0 = (int)((blockIdx.x & 7u) * (gx8 >> 3) + slot / gy) * BM;
    const int n0 = (int)(slot % gy) * BN;
    if (m0 >= a.M) return;

    const int Ktot = (MODE == 1) ? 9 * a.K : a.K;          // MODE 1: a.K = Cin
    const int nchunks = (Ktot + 2 * KP - 1) / (2 * KP);
    const int kp_total = (Ktot + 1) >> 1;

    // ---- per-thread A row bookkeeping ----
    const int a_kp = t % KP;
    int a_m[A_PER];
    int a_yx[A_PER];                                       // MODE 1: (y << 16) | x
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int r = t / KP + RPP * i;
        const int m = m0 + r;
        a_m[i] = (m < a.M) ? m : -1;
        a_yx[i] = 0;
        if (MODE == 1 && m < a.M) {
            const int hw = a.H * a.W;
            const int rem = m % hw;
            a_yx[i] = ((rem / a.W) << 16) | (rem % a.W);
        }
    }

    float2 a_reg[A_PER];
    float4 b_reg[B_PER];
    // masks of what the clamped addresses brought, applied in stage() (round 5): a mask applied where the value is loaded is a use at the point
    // of issue, and hipcc then waits for every load of the prefetch in turn before the MFMAs it is meant to hide under (yn_device.h)
    unsigned a_mk[A_PER], b_mk[B_PER];
    // 16-byte global accesses whenever the layer's channel counts / offsets are multiples of 4 floats (wave-uniform) and the
    // thread owns an even number of k-pairs
    const bool vecA = (MODE == 0) && (A_PER % 2 == 0) && ((a.K | a.in_ld | a.in_off) & 3) == 0;
    const bool vecO = ((a.N | a.out_ld | a.out_off) & 3) == 0 && (!a.pass || ((a.pass_ld | a.pass_off) & 3) == 0);

    auto prefetch = [&](int c) {
        const int k0 = c * 2 * KP;
        if (MODE == 0 && vecA) {
            // 16-byte loads: thread = (row, k-quad); a_reg[2i], a_reg[2i+1] = the quad's two k-pairs
            const int k = k0 + 4 * (t % (KP / 2));
            const bool kv = k < a.K;
#pragma unroll
            for (int i = 0; i < A_PER / 2; ++i) {
                const int m = m0 + t / (KP / 2) + (512 / KP) * i;
                const bool ok = kv && m < a.M;
                const float4 v = *reinterpret_cast<const float4*>(a.in + (size_t)(m < a.M ? m : a.M - 1) * a.in_ld + a.in_off + (kv ? k : 0));
                a_reg[2 * i] = make_float2(v.x, v.y);
                a_reg[2 * i + 1] = make_float2(v.z, v.w);
                a_mk[2 * i] = a_mk[2 * i + 1] = opaque_mask(ok);
            }
        } else if (MODE == 0) {
            const int k = k0 + 2 * a_kp;
            const bool kv = k < a.K;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const bool ok = kv && a_m[i] >= 0;
                a_reg[i] = *reinterpret_cast<const float2*>(a.in + (size_t)(a_m[i] >= 0 ? a_m[i] : a.M - 1) * a.in_ld + a.in_off + (kv ? k : 0));
                a_mk[i] = opaque_mask(ok);
            }
        } else {
            const int kk = k0 + 2 * a_kp;                   // global k = tap*Cin + ci
            const int tap = kk / a.K;
            const int ci = kk - tap * a.K;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int mc = a_m[i] >= 0 ? a_m[i] : 0;
                const int y = (a_yx[i] >> 16) + dy, x = (a_yx[i] & 0xffff) + dx;
                const bool ok = a_m[i] >= 0 && tap < 9 && y >= 0 && y < a.H && x >= 0 && x < a.W;
                const unsigned mk = opaque_mask(ok);
                const int src = ok ? mc + dy * a.W + dx : mc;          // clamped to the centre pixel when the tap is outside
                const int cic = tap < 9 ? ci : 0;
                float2 v = *reinterpret_cast<const float2*>(a.in + (size_t)src * a.in_ld + a.in_off + cic);
                a_mk[i] = mk;
                if (a.resample) {                           // (the fused add needs both values: this form masks where it loads)
                    v = vmask(v, mk);
                    const int yc = ok ? y : 0, xc = ok ? x : 0;
                    const int b = mc / (a.H * a.W);
                    size_t p2;
                    if (a.resample == 1) p2 = ((size_t)b * (a.H >> 1) + (yc >> 1)) * (a.W >> 1) + (xc >> 1);
                    else                 p2 = ((size_t)b * (a.H << 1) + (yc << 1)) * (a.W << 1) + (xc << 1);
                    const float2 u = vmask(*reinterpret_cast<const float2*>(a.in2 + p2 * a.K + cic), mk);
                    v.x += u.x; v.y += u.y;
                }
                a_reg[i] = v;
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = t + 256 * i;
            const int kp = idx / (BN / 2), c4 = idx - kp * (BN / 2);
            const int kpg = (k0 >> 1) + kp;
            const int n = n0 + c4 * 2;
            const bool ok = kpg < kp_total && n < a.Npad;
            b_reg[i] = *reinterpret_cast<const float4*>(a.Wp + ((size_t)(ok ? kpg : 0) * a.Npad + (ok ? n : 0)) * 2);
            b_mk[i] = opaque_mask(ok);
        }
    };
    auto stage = [&](int buf) {
        if (MODE == 0 && vecA) {
            const int kp = 2 * (t % (KP / 2));
#pragma unroll
            for (int i = 0; i < A_PER / 2; ++i) {
                const int r = t / (KP / 2) + (512 / KP) * i;
                *reinterpret_cast<float2*>(As + buf * KP * AS + kp * AS + r * 2) = vmask(a_reg[2 * i], a_mk[2 * i]);
                *reinterpret_cast<float2*>(As + buf * KP * AS + (kp + 1) * AS + r * 2) = vmask(a_reg[2 * i + 1], a_mk[2 * i + 1]);
            }
        } else {
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                const int r = t / KP + RPP * i;
                *reinterpret_cast<float2*>(As + buf * KP * AS + a_kp * AS + r * 2) = vmask(a_reg[i], a_mk[i]);
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = t + 256 * i;
            const int kp = idx / (BN / 2), c4 = idx - kp * (BN / 2);
            *reinterpret_cast<float4*>(Bs + buf * KP * BS + kp * BS + c4 * 4) = vmask(b_reg[i], b_mk[i]);
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    prefetch(0);
    stage(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        const int buf = (NBUF == 2) ? (c & 1) : 0;
        if (c + 1 < nchunks) prefetch(c + 1);
        const int krem = Ktot - c * 2 * KP;
        const int nq = krem >= 2 * KP ? NQ : ((krem + 3) >> 2);
        const float* Ab = As + buf * KP * AS + (wm * 32 + l31) * 2 + h * AS;
        const float* Bb = Bs + buf * KP * BS + (wn * NT * 32 + l31) * 2 + h * BS;
        // fragments of step q+1 are read from LDS before the MFMAs of step q are issued
        float2 av = *reinterpret_cast<const float2*>(Ab);
        float2 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const float2*>(Bb + nt * 64);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            if (q < nq) {                                   // wave-uniform
                float2 av_n = av, bv_n[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv_n[nt] = bv[nt];
                if (q + 1 < NQ) {
                    av_n = *reinterpret_cast<const float2*>(Ab + 2 * (q + 1) * AS);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bv_n[nt] = *reinterpret_cast<const float2*>(Bb + 2 * (q + 1) * BS + nt * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[nt].x, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[nt].y, acc[nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                av = av_n;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = bv_n[nt];
            }
        }
        if (c + 1 < nchunks) {
            if (NBUF == 1) __syncthreads();                 // every wave is done reading the only buffer
            stage(NBUF == 2 ? (buf ^ 1) : 0);
            __syncthreads();
        }
    }

    gemm_epilogue<NT>(a, acc, m0 + wm * 32, n0 + wn * NT * 32, vecO, lane);
}

// -------------------------------------------------------------------------------------------------
// Register-direct pointwise GEMM: no LDS, no barriers.  A wave owns a 32 x (32*NT) output tile for the whole K and
// feeds its MFMAs straight from global memory / L2:
//   A: lane (row r, half h) loads the 16 bytes A[r][8g+4h .. 8g+4h+3] of k-group g; two v_permlane32_swap exchange the
//      halves between lanes r and r+32, which leaves (x|y|z|w) = k (8g | 8g+1 | 8g+4 | 8g+5) in the low half and
//      (8g+2 | 8g+3 | 8g+6 | 8g+7) in the high half — exactly the operand order of gemm_conv_kernel's LDS fragments, so
//      the k-sum runs in the same order and the result is bit-identical to every tiled configuration;
//   B: lane (column c, half h) loads the float2 Wp[kp = 4g+2h (+2)][c] of the packed weights (coalesced over c).
// D k-groups are in flight per wave (loads for group g+D are issued when group g is consumed), so one wave hides the
// memory latency on its own and the kernel has no block-wide synchronisation at all: the tiled kernel parks 40 % of
// its wave-cycles in s_waitcnt/barriers on the small layers (profiles/r01_sq_counters.md).  Neighbouring waves of a
// block re-read A (WN > 1) or B (WM > 1) through L1.  Requires K, in_ld, in_off multiples of 4 floats.
// Measured (tools/direct_ablation.sh, stage-4 layer M=5408 K=N=232, 17.8 us): loads alone 19.8 us, MFMAs alone 14.7 us,
// D = 8 / 15 slower than D = 4 — the row-strided A loads (32 cache lines per instruction, 32 bytes used of each) are
// bound by L1/TA request throughput, not by latency, so this variant only edges out the LDS-tiled one (17.8 vs 19-20 us)
// on the small stage-3/4 layers, where the autotuner picks it; it is NOT the way to a higher MFMA fraction.
template <int WM, int WN, int NT, int D>
__global__ __launch_bounds__(256) void gemm_direct_kernel(GemmArgs a)
{
    constexpr int BM = 32 * WM, BN = 32 * NT * WN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    const unsigned gy = (unsigned)(a.Npad + BN - 1) / BN, gx8 = gridDim.x / gy;     // XCD-aware decode (see gemm_conv_kernel)
    const unsigned slot = blockIdx.x >> 3;
    const int m0 = (int)((blockIdx.x & 7u) * (gx8 >> 3) + slot / gy) * BM;
    const int n0 = (int)(slot % gy) * BN;
    const int mbase = m0 + wm * 32, nbase = n0 + wn * NT * 32;
    if (mbase >= a.M || nbase >= a.Npad) return;            // wave-uniform: no barriers below

    const int mr = mbase + l31;
    const float* ap = a.in + (size_t)(mr < a.M ? mr : a.M - 1) * a.in_ld + a.in_off + 4 * h;
    const int ng = (a.K + 7) >> 3;
    const float* bp[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nbase + nt * 32 + l31;                // columns >= Npad (tile wider than the matrix): load column 0, never stored
        bp[nt] = a.Wp + ((size_t)h * a.Npad + (n < a.Npad ? n : 0)) * 2;
    }
    const size_t bstep = (size_t)a.Npad * 2;                // floats per k-pair row of Wp

    float4 av[D];
    float2 bv[D][NT][2];
    auto load = [&](int g, int d) {
        // K % 8 == 4: the last group has no second k-quad.  Its loads are redirected to valid addresses (no masks: the
        // values only reach the MFMAs of that quad, which are skipped below).
        const int k = 8 * g;
        av[d] = *reinterpret_cast<const float4*>(ap + (k + 4 * h < a.K ? k : k - 4 * h));
        const size_t r0 = (size_t)(4 * g) * bstep, r1 = k + 4 < a.K ? 2 * bstep : 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            bv[d][nt][0] = *reinterpret_cast<const float2*>(bp[nt] + r0);
            bv[d][nt][1] = *reinterpret_cast<const float2*>(bp[nt] + r0 + r1);
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

#pragma unroll
    for (int d = 0; d < D; ++d)
        if (d < ng) load(d, d);
    for (int g0 = 0; g0 < ng; g0 += D) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const int g = g0 + d;
            if (g < ng) {                                    // wave-uniform
                float4 x = av[d];
                float2 b[NT][2];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) { b[nt][0] = bv[d][nt][0]; b[nt][1] = bv[d][nt][1]; }
#ifndef YN_EXP_NO_LOAD
                if (g + D < ng) load(g + D, d);
#endif
                {   // (x,z) and (y,w): swap the high half of the first with the low half of the second
                    const auto s0 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.x), __float_as_uint(x.z), false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(__float_as_uint(x.y), __float_as_uint(x.w), false, false);
                    x = make_float4(__uint_as_float(s0[0]), __uint_as_float(s1[0]), __uint_as_float(s0[1]), __uint_as_float(s1[1]));
                }
                __builtin_amdgcn_sched_barrier(0);
#ifdef YN_EXP_NO_MFMA
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt][0] += x.x * b[nt][0].x + x.y * b[nt][0].y + x.z * b[nt][1].x + x.w * b[nt][1].y;
#else
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.x, b[nt][0].x, acc[nt], 0, 0, 0);
                    acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.y, b[nt][0].y, acc[nt], 0, 0, 0);
                }
                if (8 * g + 4 < a.K) {                       // wave-uniform: the second k-quad of the group exists
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.z, b[nt][1].x, acc[nt], 0, 0, 0);
                        acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(x.w, b[nt][1].y, acc[nt], 0, 0, 0);
                    }
                }
#endif
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    const bool vecO = ((a.N | a.out_ld | a.out_off) & 3) == 0 && (!a.pass || ((a.pass_ld | a.pass_off) & 3) == 0);
    gemm_epilogue<NT>(a, acc, mbase, nbase, vecO, lane);
}

// -------------------------------------------------------------------------------------------------
// Dense 3x3 (stride 1, pad 1) on the MFMA with the input tile resident in LDS.
// A block owns 128 consecutive output pixels (flat NHW order) and stages, ONCE, the flat pixel range
// [p0 - W - 1, p0 + 128 + W + 1) x Cin — with the FPN/PAN `+ up2(x2)` / `+ down(x2)` add fused in — into
// LDS.  All 9 taps x Cin/32 K-chunks then read their A fragments straight from that halo (the tap is an
// address offset; image-border taps are predicated to zero per lane), so the K loop only streams the
// packed weights (double-buffered).  Pixel stride Cin+2 floats makes the fragment ds_read_b64 conflict-free.
// -------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(256) void conv3x3_halo_kernel(GemmArgs a)
{
    constexpr int BM = 128, BN = 32 * NT, KP = 16, BS = BN * 2, B_PER = BN / 32;
    extern __shared__ __attribute__((aligned(16))) float c3_smem[];
    const int Cin = a.K, CS = Cin + 2;
    const int W = a.W, H = a.H, HW = H * W;
    const int npix = BM + 2 * W + 2;
    float* halo = c3_smem;                                   // [npix][CS]
    float* Bs = c3_smem + ((npix * CS + 3) & ~3);            // [2][KP][BS]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int p0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;
    const int base = p0 - W - 1;
    const int cpt = Cin >> 5;                               // K-chunks per tap
    const int nchunks = 9 * cpt;
    const int kp_total = (9 * Cin) >> 1;

#ifdef YN_EXP_TIMING
    const long long T0 = __builtin_readcyclecounter();
#endif
    float4 b_reg[B_PER];
    auto prefetch_b = [&](int c) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = t + 256 * i;
            const int kp = idx / (BN / 2), c4 = idx - kp * (BN / 2);
            const int kpg = c * KP + kp;
            const int n = n0 + c4 * 2;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (kpg < kp_total && n < a.Npad) v = *reinterpret_cast<const float4*>(a.Wp + ((size_t)kpg * a.Npad + n) * 2);
            b_reg[i] = v;
        }
    };
    auto stage_b = [&](int buf) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = t + 256 * i;
            const int kp = idx / (BN / 2), c4 = idx - kp * (BN / 2);
            *reinterpret_cast<float4*>(Bs + buf * KP * BS + kp * BS + c4 * 4) = b_reg[i];
        }
    };
    prefetch_b(0);

    // ---- stage the halo (fused resample-add), thread = (channel pair, pixel lane) ----
    // U pixels per batch: all global loads of a batch are issued before the first LDS store waits on them.
    {
        constexpr int U = 8;
        const int cpn = Cin >> 1;
        const int ppl = 256 / cpn;                           // pixels per pass
        const int cp = t % cpn, pl = t / cpn;
        if (pl < ppl) {
            for (int i0 = pl; i0 < npix; i0 += ppl * U) {
                float2 v[U], u2[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * ppl;
                    const int q = base + i;
                    v[u] = make_float2(0.0f, 0.0f);
                    u2[u] = make_float2(0.0f, 0.0f);
                    if (i < npix && q >= 0 && q < a.M) {
                        v[u] = *reinterpret_cast<const float2*>(a.in + (size_t)q * a.in_ld + a.in_off + 2 * cp);
                        if (a.resample) {
                            const int b = q / HW, rem = q - b * HW;
                            const int y = rem / W, x = rem - y * W;
                            size_t p2;
                            if (a.resample == 1) p2 = ((size_t)b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
                            else                 p2 = ((size_t)b * (H << 1) + (y << 1)) * (W << 1) + (x << 1);
                            u2[u] = *reinterpret_cast<const float2*>(a.in2 + p2 * Cin + 2 * cp);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * ppl;
                    if (i < npix) *reinterpret_cast<float2*>(halo + i * CS + 2 * cp) = make_float2(v[u].x + u2[u].x, v[u].y + u2[u].y);
                }
            }
        }
    }
    stage_b(0);

    // ---- per-lane tap validity of this lane's output pixel ----
    const int r = wave * 32 + l31;                           // row of the tile owned by this lane
    const int m = p0 + r;
    unsigned tapmask = 0;
    if (m < a.M) {
        const int rem = m % HW;
        const int y = rem / W, x = rem - y * W;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) tapmask |= 1u << tap;
        }
    }
    __syncthreads();

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.0f;
#ifdef YN_EXP_TIMING
    const long long T1 = __builtin_readcyclecounter();
#endif

    for (int c = 0; c < nchunks; ++c) {
        const int buf = c & 1;
#ifndef YN_EXP_NOSTAGE
        if (c + 1 < nchunks) prefetch_b(c + 1);
#endif
        const int tap = c / cpt;
        const int ck = (c - tap * cpt) << 5;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const bool ok = (tapmask >> tap) & 1u;
        const float* Ab = halo + (W + 1 + r + dy * W + dx) * CS + ck + 2 * h;
        const float* Bb = Bs + buf * KP * BS + l31 * 2;
        // fragments of step q+1 are read from LDS before the MFMAs of step q are issued, so the ds_read
        // latency hides behind 2*NT MFMAs (the compiler otherwise sinks each read next to its first use)
        float2 av = *reinterpret_cast<const float2*>(Ab);
        float2 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const float2*>(Bb + h * BS + nt * 64);
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            float2 av_n = av, bv_n[NT];
            if (q + 1 < 8) {
                av_n = *reinterpret_cast<const float2*>(Ab + 4 * (q + 1));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv_n[nt] = *reinterpret_cast<const float2*>(Bb + (2 * (q + 1) + h) * BS + nt * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float ax = ok ? av.x : 0.0f, ay = ok ? av.y : 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bv[nt].x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, bv[nt].y, acc[nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (q + 1 < 8) {
                av = av_n;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = bv_n[nt];
            }
        }
#ifndef YN_EXP_NOSTAGE
        if (c + 1 < nchunks) stage_b(buf ^ 1);
        __syncthreads();
#endif
    }
#ifdef YN_EXP_TIMING
    const long long T2 = __builtin_readcyclecounter();
    if (blockIdx.x == 1 && blockIdx.y == 0 && (t & 63) == 0)
        printf("c3 W=%d wave %d: prologue %lld  loop %lld (%lld/chunk)  [cycles]\n", W, wave, T1 - T0, T2 - T1, (T2 - T1) / nchunks);
#endif

#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + nt * 32 + l31;
        if (n >= a.N) continue;
        const float bias = a.bias[n];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
            const int mm = p0 + wave * 32 + row;
            if (mm >= a.M) continue;
            a.out[(size_t)mm * a.out_ld + a.out_off + n] = apply_act(acc[nt][k] + bias, a.act);
        }
    }
}

// Same algorithm, specialised for a compile-time Cin: the weight stream is chunked per TAP (Cin k-values, one
// LDS buffer, next tap prefetched into registers) => 9 chunk boundaries instead of 9*Cin/32, and the halo is
// staged with 16-byte global loads.
template <int NT, int CIN, int SPLIT>
__global__ __launch_bounds__(256) void conv3x3_halo_tap_kernel(GemmArgs a)
{
    constexpr int BM = 128, BN = 32 * NT, BS = BN * 2, KPT = CIN / 2 / SPLIT;   // k-pairs per weight chunk (a tap, or 1/SPLIT of one)
    constexpr int B_PER = (KPT * BN / 2 + 255) / 256;                      // float4 per thread per tap
    constexpr int CS = CIN + 2, NQ = CIN / 4;
    extern __shared__ __attribute__((aligned(16))) float c3t_smem[];
    const int W = a.W, H = a.H, HW = H * W;
    const int npix = BM + 2 * W + 2;
    float* halo = c3t_smem;                                  // [npix][CS]
    float* Bs = c3t_smem + ((npix * CS + 3) & ~3);           // [KPT][BS]
    constexpr int NQ_C = KPT / 2;                            // MFMA k-steps (of 4 k) per chunk

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int p0 = (int)xcd_block(blockIdx.x, gridDim.x) * BM;   // XCD-contiguous tile order: neighbouring tiles share their halo in ONE L2
    if (p0 >= a.M) return;                                       // gridDim.x is rounded up to a multiple of 8
    const int n0 = blockIdx.y * BN;
    const int base = p0 - W - 1;

    float4 b_reg[B_PER];
    auto prefetch_b = [&](int tap) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = t + 256 * i;
            const int kp = idx / (BN / 2), c4 = idx - kp * (BN / 2);
            const int n = n0 + c4 * 2;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (kp < KPT && n < a.Npad) v = *reinterpret_cast<const float4*>(a.Wp + ((size_t)(tap * KPT + kp) * a.Npad + n) * 2);
            b_reg[i] = v;
        }
    };
    auto stage_b = [&]() {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = t + 256 * i;
            const int kp = idx / (BN / 2), c4 = idx - kp * (BN / 2);
            if (kp < KPT) *reinterpret_cast<float4*>(Bs + kp * BS + c4 * 4) = b_reg[i];
        }
    };
    prefetch_b(0);

    // ---- halo, 16-byte loads: thread = (channel quad, pixel lane) ----
    {
        constexpr int U = 8, CQ = CIN / 4, PPL = 256 / CQ;
        const int cq = t % CQ, pl = t / CQ;
        if (pl < PPL) {
            for (int i0 = pl; i0 < npix; i0 += PPL * U) {
                float4 v[U], u2[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * PPL;
                    const int q = base + i;
                    v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    u2[u] = v[u];
                    if (i < npix && q >= 0 && q < a.M) {     // (measured: the unconditional masked form is 10 % SLOWER here)
                        v[u] = *reinterpret_cast<const float4*>(a.in + (size_t)q * CIN + 4 * cq);
                        if (a.resample) {
                            const int b = q / HW, rem = q - b * HW;
                            const int y = rem / W, x = rem - y * W;
                            size_t p2;
                            if (a.resample == 1) p2 = ((size_t)b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
                            else                 p2 = ((size_t)b * (H << 1) + (y << 1)) * (W << 1) + (x << 1);
                            u2[u] = *reinterpret_cast<const float4*>(a.in2 + p2 * CIN + 4 * cq);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * PPL;
                    if (i < npix) {                          // pixel stride CS*4 bytes is only 8-byte aligned
                        *reinterpret_cast<float2*>(halo + i * CS + 4 * cq) = make_float2(v[u].x + u2[u].x, v[u].y + u2[u].y);
                        *reinterpret_cast<float2*>(halo + i * CS + 4 * cq + 2) = make_float2(v[u].z + u2[u].z, v[u].w + u2[u].w);
                    }
                }
            }
        }
    }
    stage_b();

    const int r = wave * 32 + l31;
    const int m = p0 + r;
    unsigned tapmask = 0;
    if (m < a.M) {
        const int rem = m % HW;
        const int y = rem / W, x = rem - y * W;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) tapmask |= 1u << tap;
        }
    }
    __syncthreads();

    f32x16 acc[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[i][k] = 0.0f;

    for (int chunk = 0; chunk < 9 * SPLIT; ++chunk) {
        if (chunk + 1 < 9 * SPLIT) prefetch_b(chunk + 1);
        const int tap = chunk / SPLIT, part = chunk - tap * SPLIT;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const bool ok = (tapmask >> tap) & 1u;
        const float* Ab = halo + (W + 1 + r + dy * W + dx) * CS + 2 * h + part * (CIN / SPLIT);
        const float* Bb = Bs + l31 * 2 + h * BS;
        float2 av = *reinterpret_cast<const float2*>(Ab);
        float2 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const float2*>(Bb + nt * 64);
#pragma unroll 8
        for (int q = 0; q < NQ_C; ++q) {
            float2 av_n = av, bv_n[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv_n[nt] = bv[nt];
            if (q + 1 < NQ_C) {
                av_n = *reinterpret_cast<const float2*>(Ab + 4 * (q + 1));
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv_n[nt] = *reinterpret_cast<const float2*>(Bb + 2 * (q + 1) * BS + nt * 64);
            }
            __builtin_amdgcn_sched_barrier(0);
            const float ax = ok ? av.x : 0.0f, ay = ok ? av.y : 0.0f;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ax, bv[nt].x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(ay, bv[nt].y, acc[nt], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            av = av_n;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = bv_n[nt];
        }
        if (chunk + 1 < 9 * SPLIT) {
            __syncthreads();                                 // everyone is done with this chunk's weights
            stage_b();
            __syncthreads();
        }
    }

    // epilogue: quad transpose -> 16-byte stores (out_ld = N-stride, multiples of 4 here)
    const int j = lane & 3;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ncol = n0 + nt * 32 + l31;
        const float bias = ncol < a.N ? a.bias[ncol] : 0.0f;
        const int nq = n0 + nt * 32 + (l31 & ~3);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v0 = apply_act(acc[nt][4 * g + 0] + bias, a.act), v1 = apply_act(acc[nt][4 * g + 1] + bias, a.act);
            float v2 = apply_act(acc[nt][4 * g + 2] + bias, a.act), v3 = apply_act(acc[nt][4 * g + 3] + bias, a.act);
            {
                const float s0 = (j & 1) ? v0 : v1, s1 = (j & 1) ? v2 : v3;
                const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
                if (j & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
            }
            {
                const float s0 = (j & 2) ? v0 : v2, s1 = (j & 2) ? v1 : v3;
                const float r0 = quad_xor2(s0), r1 = quad_xor2(s1);
                if (j & 2) { v0 = r0; v1 = r1; } else { v2 = r0; v3 = r1; }
            }
            const int mm = p0 + wave * 32 + 8 * g + 4 * h + j;
            if (mm < a.M && nq < a.N)
                *reinterpret_cast<float4*>(a.out + (size_t)mm * a.out_ld + a.out_off + nq) = make_float4(v0, v1, v2, v3);
        }
    }
}

// -------------------------------------------------------------------------------------------------
// The 96 -> 96 neck convolutions on the F16 matrix pipe with SPLIT operands: fp32-class results at 4-5x the f32 MFMA rate.
// gfx950 has no TF32/xf32 (MI355X_MICROARCH.md); its f32-input MFMA runs at 1/16 of the f16 rate, and the dense 3x3 layers are the
// MFMA-bound part of the network (36 % of the flops, 0.5 of the f32 peak with conv3x3_halo_tap_kernel).  Every fp32 operand is
// written as x = hi + lo * 2^-11 with hi = f16(x), lo = f16((x - hi) * 2^11): x - hi is exact in fp32, the scaling keeps lo out of
// the f16 subnormals, and |x - (hi + lo * 2^-11)| <= 2^-22 |x|.  Then a*w = ah*wh + (ah*wl + al*wh) * 2^-11 + O(2^-22): three f16
// MFMAs (products exact in the fp32 accumulators) into TWO accumulator sets, combined once in the epilogue.  Per-product error
// <= ~3 * 2^-22 relative — the size of the fp32 path's own accumulation rounding; measured against the float64 oracle the two paths
// are indistinguishable (tests/test_gpu_parity.py::test_split_f16_conv_is_fp32_class).  Operands must fit the f16 range (|x| < 65504:
// normalised activations and folded weights are O(1)); YN_EXACT_F32=1 keeps the f32-MFMA kernel.
// Structure as conv3x3_halo_tap_kernel: the pixel range of a flat 128-pixel tile + halo is staged ONCE in LDS (already split, two
// planes of halves, FPN/PAN resample-add fused), the nine taps read their A fragments from it, the pre-split packed weights stream
// per tap through one LDS buffer with register prefetch.
// -------------------------------------------------------------------------------------------------

// The 96 input channels go through LDS in two HALVES of 48 (halo planes of 48 channels, then the nine taps on them, twice): 71 KB
// per workgroup instead of 134 KB, so TWO workgroups share a CU and one's halo phase (HBM-bound) and epilogue overlap the other's tap
// phase (LDS / MFMA-bound) — with one resident workgroup the three phases ran strictly one after the other.
// NH = 1 (all 96 channels staged at once) is kept for launches of fewer than 256 workgroups (13x13 maps, bs = 1): nothing to overlap
// with there, and a second halo phase costs 3 us.  Both forms walk K in the same order — channel half, tap, 16-deep step — so a
// layer's bits do not depend on which one its size selects (an image's detections are independent of its batch:
// test_infer_config2_bs32).
// TPS = taps of weights staged per step (round 5).  A step costs a barrier pair and a weight fetch that was requested one step earlier - ~1.2 us
// whatever it multiplies - and the small-map forms (NT = 1: 6 KB of weights per tap and channel half) are nothing but 18 of those: smooth_0 (5 408
// pixels) took as long as smooth_2 (21 632).  TPS = 3 / 9 makes that 6 / 2 steps (18 / 54 KB of LDS); same (half, tap, k-step) order: same bits.
template <int NT, int NH, int TPS = 1>
__global__ __launch_bounds__(256, NH) void conv3x3_split_kernel(GemmArgs a)
{
    constexpr int CIN = 96, CH = 48, HC = CIN / NH, KQ = CH / 8, KQT = CIN / 8, BM = 128, BN = 32 * NT, CSH = HC + 8;   // halo row stride (halves): 112 / 208 bytes
    constexpr int WCH = KQ * BN * 8;                                                  // halves per weight plane per (tap, half)
    constexpr int B_PER = (2 * KQ * BN + 255) / 256;                                  // 16-byte granules per thread per (tap, half) (hi and lo planes)
    constexpr int NSTEP = 18 / TPS;
    static_assert(9 % TPS == 0, "a step stays inside one channel half");
    extern __shared__ __attribute__((aligned(16))) float c3s_smem[];
    const int W = a.W, H = a.H, HW = H * W;
    const int npix = BM + 2 * W + 2;
    c3h16* Hh = reinterpret_cast<c3h16*>(c3s_smem);                                   // [npix][CSH]
    c3h16* Hl = Hh + (size_t)npix * CSH;
    c3h16* Bh = Hl + (size_t)npix * CSH;                                              // TPS x { [KQ][BN][8] hi, then lo }
    c3h16* Bl = Bh + WCH;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int p0 = (int)xcd_block(blockIdx.x, gridDim.x) * BM;
    if (p0 >= a.M) return;
    const int n0 = blockIdx.y * BN;
    const int base = p0 - W - 1;
    const c3h16* Wsh = reinterpret_cast<const c3h16*>(a.Wsh);
    const c3h16* Wsl = reinterpret_cast<const c3h16*>(a.Wsl);

#ifdef YN_EXP_TIMING
    long long TS[8]; int tsn = 0;
#define YN_TS() TS[tsn++] = __builtin_readcyclecounter()
#else
#define YN_TS()
#endif
    YN_TS();
    c3h16x8 b_reg[TPS][B_PER];
    auto prefetch_b = [&](int step) {                                                 // step = (half * 9 + first tap) / TPS
        const int half = (step * TPS) / 9, tap0 = step * TPS - half * 9;
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt)
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int g = t + 256 * i;                                            // granule: plane (hi / lo), octet o, column n
                const int pl = g / (KQ * BN), r = g - pl * (KQ * BN);
                const int o = r / BN, n = r - o * BN;
                c3h16x8 v;
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (c3h16)0.0f;
                if (g < 2 * KQ * BN) v = *reinterpret_cast<const c3h16x8*>((pl ? Wsl : Wsh) + (((size_t)(tap0 + tt) * KQT + half * KQ + o) * a.Npad + n0 + n) * 8);
                b_reg[tt][i] = v;
            }
    };
    auto stage_b = [&]() {
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt)
#pragma unroll
            for (int i = 0; i < B_PER; ++i) {
                const int g = t + 256 * i;
                if (g < 2 * KQ * BN) *reinterpret_cast<c3h16x8*>(Bh + (size_t)tt * 2 * WCH + (size_t)g * 8) = b_reg[tt][i];      // Bl follows Bh: plane 1 lands there
            }
    };
    prefetch_b(0);

    // ---- halo: 16-byte global loads (+ the fused resample-add), split, two 8-byte LDS stores per item.  Phase timing (tools/phase_timing.sh c3split,
    //      smooth_1: W = 52, one block per CU): halo 19 k cycles, nine taps 31.7 k (486 MFMAs = 15.5 k), epilogue 9 k per 128-pixel tile.
    //      The halo phase is BANDWIDTH-bound, not latency-bound: 180 KB per tile (halo factor 1.83 at W = 52 + the up2 source) at the
    //      ~10 B/clk/CU every CU gets when all of them stream at once; 24 loads in flight per thread instead of 8 changed nothing ----
    float amax = 0.0f;                                                                // range guard (yn_device.h): largest |value| this thread has split
    auto load_halo = [&](int half) {
        constexpr int U = 8, CQ = HC / 4, PPL = 256 / CQ;
        const int cq = t % CQ, pl = t / CQ;
        const int c0 = half * HC + 4 * cq;
        if (pl < PPL) {
            for (int i0 = pl; i0 < npix; i0 += PPL * U) {
                float4 v[U], u2[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * PPL;
                    const int q = base + i;
                    v[u] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    u2[u] = v[u];
                    if (i < npix && q >= 0 && q < a.M) {
                        v[u] = *reinterpret_cast<const float4*>(a.in + (size_t)q * CIN + c0);
                        if (a.resample) {
                            const int b = q / HW, rem = q - b * HW;
                            const int y = rem / W, x = rem - y * W;
                            size_t p2;
                            if (a.resample == 1) p2 = ((size_t)b * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1);
                            else                 p2 = ((size_t)b * (H << 1) + (y << 1)) * (W << 1) + (x << 1);
                            u2[u] = *reinterpret_cast<const float4*>(a.in2 + p2 * CIN + c0);
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = i0 + u * PPL;
                    if (i < npix) {
                        const float x4[4] = {v[u].x + u2[u].x, v[u].y + u2[u].y, v[u].z + u2[u].z, v[u].w + u2[u].w};
                        c3h16x4 hi, lo;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { amax = range_track(amax, x4[j]); hi[j] = (c3h16)x4[j]; lo[j] = (c3h16)((x4[j] - (float)hi[j]) * 2048.0f); }
                        *reinterpret_cast<c3h16x4*>(Hh + (size_t)i * CSH + 4 * cq) = hi;
                        *reinterpret_cast<c3h16x4*>(Hl + (size_t)i * CSH + 4 * cq) = lo;
                    }
                }
            }
        }
    };
    load_halo(0);
    YN_TS();
    stage_b();

    const int r = wave * 32 + l31;
    const int m = p0 + r;
    unsigned tapmask = 0;
    if (m < a.M) {
        const int rem = m % HW;
        const int y = rem / W, x = rem - y * W;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) tapmask |= 1u << tap;
        }
    }
    __syncthreads();
    YN_TS();

    f32x16 acc0[NT], acc1[NT];                              // sum ah*wh ; sum (ah*wl + al*wh), worth 2^-11
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc0[i][k] = 0.0f; acc1[i][k] = 0.0f; }

    for (int step = 0; step < NSTEP; ++step) {              // (channel half, TPS taps): K = 2 x 9 x 48
        if (step + 1 < NSTEP) prefetch_b(step + 1);
#pragma unroll
        for (int tt = 0; tt < TPS; ++tt) {
        const int tap = (step * TPS) % 9 + tt;
        const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const bool ok = (tapmask >> tap) & 1u;
        const size_t arow = (size_t)(W + 1 + r + dy * W + dx) * CSH + h * 8 + (NH == 1 ? ((step * TPS) / 9) * CH : 0);
        const c3h16* Bhb = Bh + (size_t)tt * 2 * WCH + (size_t)(h * BN + l31) * 8;
        const c3h16* Blb = Bl + (size_t)tt * 2 * WCH + (size_t)(h * BN + l31) * 8;
#pragma unroll
        for (int ks = 0; ks < KQ / 2; ++ks) {
            c3h16x8 ah = *reinterpret_cast<const c3h16x8*>(Hh + arow + ks * 16);
            c3h16x8 al = *reinterpret_cast<const c3h16x8*>(Hl + arow + ks * 16);
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { ah[j] = (c3h16)0.0f; al[j] = (c3h16)0.0f; }
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const c3h16x8 bh = *reinterpret_cast<const c3h16x8*>(Bhb + (size_t)(ks * 2 * BN + nt * 32) * 8);
                const c3h16x8 bl = *reinterpret_cast<const c3h16x8*>(Blb + (size_t)(ks * 2 * BN + nt * 32) * 8);
                acc0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0[nt], 0, 0, 0);
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[nt], 0, 0, 0);
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[nt], 0, 0, 0);
            }
        }
        }
        if (step + 1 < NSTEP) {
            __syncthreads();                                 // everyone is done with this step's weights (and, after tap 8, with the halo planes)
            if (NH == 2 && (step * TPS) % 9 + TPS == 9) load_halo(1);
            stage_b();
            __syncthreads();
        }
    }

    YN_TS();
    range_report(a.ovf, amax);
    // epilogue: combine the two accumulator sets, bias + activation, quad transpose -> 16-byte stores
    const int j = lane & 3;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int ncol = n0 + nt * 32 + l31;
        const float bias = ncol < a.N ? a.bias[ncol] : 0.0f;
        const int nq = n0 + nt * 32 + (l31 & ~3);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float vv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) vv[e] = apply_act(__builtin_fmaf(acc1[nt][4 * g + e], 1.0f / 2048.0f, acc0[nt][4 * g + e]) + bias, a.act);
            float v0 = vv[0], v1 = vv[1], v2 = vv[2], v3 = vv[3];
            {
                const float s0 = (j & 1) ? v0 : v1, s1 = (j & 1) ? v2 : v3;
                const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
                if (j & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
            }
            {
                const float s0 = (j & 2) ? v0 : v2, s1 = (j & 2) ? v1 : v3;
                const float r0 = quad_xor2(s0), r1 = quad_xor2(s1);
                if (j & 2) { v0 = r0; v1 = r1; } else { v2 = r0; v3 = r1; }
            }
            const int mm = p0 + wave * 32 + 8 * g + 4 * h + j;
            if (mm < a.M && nq < a.N)
                *reinterpret_cast<float4*>(a.out + (size_t)mm * a.out_ld + a.out_off + nq) = make_float4(v0, v1, v2, v3);
        }
    }
#ifdef YN_EXP_TIMING
    YN_TS();
    if (t == 0 && (blockIdx.x % 61) == 7 && blockIdx.y == 0)
        printf("c3split NT %d blk %d start %lld halo %lld stageb+sync %lld taps %lld epi %lld\n", NT, (int)blockIdx.x, TS[0], TS[1] - TS[0], TS[2] - TS[1], TS[3] - TS[2], TS[4] - TS[3]);
#endif
#undef YN_TS
}
static size_t conv3x3_split_lds(int W, int NT, int NH, int TPS = 1) { return ((size_t)2 * (128 + 2 * W + 2) * (96 / NH + 8) + (size_t)TPS * 2 * 6 * (32 * NT) * 8) * 2; }

static size_t conv3x3_halo_tap_lds(int W, int Cin, int NT, int split = 1)
{
    const int npix = 128 + 2 * W + 2;
    return ((size_t)((npix * (Cin + 2) + 3) & ~3) + (size_t)(Cin / 2 / split) * (32 * NT * 2)) * sizeof(float);
}

static size_t conv3x3_halo_lds(int W, int Cin, int NT)
{
    const int npix = 128 + 2 * W + 2;
    return ((size_t)((npix * (Cin + 2) + 3) & ~3) + 2 * 16 * (32 * NT * 2)) * sizeof(float);
}

// -------------------------------------------------------------------------------------------------
// Pointwise (1x1) convolution on the f16 matrix pipe with split fp32 operands — the scheme of conv3x3_split_kernel (x = hi + lo*2^-11,
// three f16 MFMAs per product into two fp32 accumulator sets) for the GEMMs of the network, whose f32-MFMA form spends as long in
// its MFMA phase as in its memory phase (profiles/r01_sq_counters.md: 50 % issue-stalled on the matrix pipe).  Block = 4 waves as
// WM x WN, 32 x (32*NT) per wave; K walks in chunks of 32: the fp32 A rows are loaded coalesced (16- or 8-byte accesses), split in
// registers and staged as two planes of halves, the pre-split packed weights are copied straight; next chunk's global loads fly
// during the MFMAs.  Epilogue = gemm_epilogue (bias, activation, concat+shuffle interleave, 16-byte stores).  Every configuration
// runs the same MFMA sequence per k-chunk in the same chunk order: all split configurations are bit-identical to each other (not
// to the f32-MFMA family, which rounds once per product-add; against float64 the split form is the more accurate of the two).
// -------------------------------------------------------------------------------------------------
template <int WM, int WN, int NT, int KC>
__device__ __forceinline__ void gemm_split_block(const GemmArgs& a, c3h16* smem, unsigned bid, unsigned nblocks)
{
    constexpr int BM = 32 * WM, BN = 32 * NT * WN;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int wm = wave % WM, wn = wave / WM;
    const unsigned gy = (unsigned)(a.Npad + BN - 1) / BN, gx8 = nblocks / gy;     // XCD-aware decode (see gemm_conv_kernel)
    const unsigned slot = bid >> 3;
    const int m0 = (int)((bid & 7u) * (gx8 >> 3) + slot / gy) * BM;
    const int n0 = (int)(slot % gy) * BN;
    if (m0 >= a.M) return;
    const bool vecO = ((a.N | a.out_ld | a.out_off) & 3) == 0 && (!a.pass || ((a.pass_ld | a.pass_off) & 3) == 0);
    f32x16 acc0[NT];
    gemm_split_tile<WM, WN, NT, KC>(a, smem, m0, n0, acc0);
    gemm_epilogue<NT>(a, acc0, m0 + wm * 32, n0 + wn * NT * 32, vecO, lane);
}

template <int WM, int WN, int NT, int KC>
__global__ __launch_bounds__(256) void gemm_split_kernel(GemmArgs a)
{
    __shared__ __attribute__((aligned(16))) c3h16 smem[gemm_split_smem_halves(32 * WM, 32 * NT * WN, KC)];
    gemm_split_block<WM, WN, NT, KC>(a, smem, blockIdx.x, gridDim.x);
}

template <int WM, int WN, int NT, int KC>
__global__ __launch_bounds__(256) void gemm_split_group_kernel(Group<GemmArgs> g)
{
    __shared__ __attribute__((aligned(16))) c3h16 smem[gemm_split_smem_halves(32 * WM, 32 * NT * WN, KC)];
    unsigned local, nb;
    const int p = group_problem(g.first, blockIdx.x, local, nb);
    gemm_split_block<WM, WN, NT, KC>(g.a[p], smem, local, nb);
}

struct TileCfg { int WM, WN, NT, KP, NBUF; };

// Every instantiated tile configuration of the pointwise GEMM.  All of them compute bit-identical results,
// so the per-layer choice (heuristic below, or the handle's autotuner via GemmArgs::cfg) is purely a speed matter.
#define YN_PW_CONFIGS(X)                                                                             \
    X(4, 1, 1, 16, 2) X(4, 1, 2, 16, 2) X(4, 1, 3, 16, 2) X(4, 1, 4, 16, 2) X(4, 1, 8, 16, 2)        \
    X(2, 2, 1, 16, 2) X(2, 2, 2, 16, 2) X(2, 2, 4, 16, 2) X(1, 4, 1, 16, 2) X(1, 4, 2, 16, 2)        \
    X(4, 1, 1, 32, 1) X(4, 1, 2, 32, 1) X(4, 1, 3, 32, 1) X(4, 1, 4, 32, 1)                          \
    X(2, 2, 1, 32, 1) X(2, 2, 2, 32, 1) X(1, 4, 1, 32, 1) X(1, 4, 2, 32, 1)                          \
    X(2, 2, 1, 32, 2) X(2, 2, 2, 32, 2) X(1, 4, 1, 32, 2) X(4, 1, 1, 32, 2) X(4, 1, 2, 32, 2)        \
    X(1, 4, 1, 16, 1) X(2, 2, 1, 16, 1) X(4, 1, 1, 16, 1) X(1, 4, 2, 16, 1) X(4, 1, 3, 16, 1)        \
    X(2, 2, 1, 8, 1) X(2, 2, 2, 8, 1) X(4, 1, 2, 8, 1) X(4, 1, 4, 8, 1) X(2, 2, 4, 8, 1) X(1, 4, 1, 8, 1)

static const TileCfg g_pw_cfgs[] = {
#define X(wm, wn, nt, kp, nb) {wm, wn, nt, kp, nb},
    YN_PW_CONFIGS(X)
#undef X
};
static const char* const g_pw_names[] = {
#define X(wm, wn, nt, kp, nb) "gemm_conv_kernel<" #wm "," #wn "," #nt ",0," #kp "," #nb ">",
    YN_PW_CONFIGS(X)
#undef X
};
constexpr int N_PW_CFGS = (int)(sizeof(g_pw_cfgs) / sizeof(g_pw_cfgs[0]));

// register-direct configurations (gemm_direct_kernel<WM, WN, NT, D>): bit-identical to the tiled ones as well
#define YN_PWD_CONFIGS(X)                                                                            \
    X(4, 1, 1, 4) X(2, 2, 1, 4) X(1, 4, 1, 4) X(4, 1, 2, 4) X(2, 2, 2, 4) X(4, 1, 3, 4) X(4, 1, 4, 3) X(2, 2, 4, 3)
struct DirectCfg { int WM, WN, NT, D; };
static const DirectCfg g_pwd_cfgs[] = {
#define X(wm, wn, nt, d) {wm, wn, nt, d},
    YN_PWD_CONFIGS(X)
#undef X
};
static const char* const g_pwd_names[] = {
#define X(wm, wn, nt, d) "gemm_direct_kernel<" #wm "," #wn "," #nt "," #d ">",
    YN_PWD_CONFIGS(X)
#undef X
};
constexpr int N_PWD_CFGS = (int)(sizeof(g_pwd_cfgs) / sizeof(g_pwd_cfgs[0]));

// split-f16 configurations (gemm_split_kernel<WM, WN, NT, KC>): bit-identical to each other; used when the layer carries split packs.
// KC = 64: half as many barrier rounds for the layers that are a chain of them (small M, K = 232 / 464); only picked by the autotuner.
#define YN_PWS_CONFIGS(X) X(4, 1, 1, 32) X(4, 1, 2, 32) X(4, 1, 3, 32) X(4, 1, 4, 32) X(2, 2, 1, 32) X(2, 2, 2, 32) X(1, 4, 1, 32) X(1, 4, 2, 32) X(2, 2, 4, 32) \
    X(2, 2, 1, 64) X(2, 2, 2, 64) X(1, 4, 1, 64) X(4, 1, 1, 64) X(1, 4, 1, 128)
struct SplitCfg { int WM, WN, NT, KC; };
static const SplitCfg g_pws_cfgs[] = {
#define X(wm, wn, nt, kc) {wm, wn, nt, kc},
    YN_PWS_CONFIGS(X)
#undef X
};
static const char* const g_pws_names[] = {
#define X(wm, wn, nt, kc) "gemm_split_kernel<" #wm "," #wn "," #nt "," #kc ">",
    YN_PWS_CONFIGS(X)
#undef X
};
constexpr int N_PWS_CFGS = (int)(sizeof(g_pws_cfgs) / sizeof(g_pws_cfgs[0]));

// tile configurations of gemm_conv_kernel, then the register-direct configurations (the f32-MFMA family), then the split-f16 family
int pw_config_count() { return N_PW_CFGS + N_PWD_CFGS + N_PWS_CFGS + 1; }       // + pw_pipe_kernel (kernels_pipe.hip)
int pw_f32_config_count() { return N_PW_CFGS + N_PWD_CFGS; }

static bool launch_pw_split(const GemmArgs& a, int idx, hipStream_t s)
{
    if (!a.Wsh || !a.Wsl || (a.K & 1) || (a.in_ld & 1) || (a.in_off & 1)) return false;
    if (idx < 0 || idx >= N_PWS_CFGS) {                     // heuristic: widest tile that still gives >= 2 blocks per CU
        const int nt32 = a.Npad / 32;
        idx = 0;
        long best = -1;
        for (int i = 0; i < N_PWS_CFGS; ++i) {
            const SplitCfg& c = g_pws_cfgs[i];
            if (c.KC != 32) continue;
            const long BM = 32 * c.WM, BN = 32 * c.NT * c.WN;
            if (BN > 32 * nt32 + 31 && BN != 32) continue;
            const long blocks = ((a.M + BM - 1) / BM) * ((a.Npad + BN - 1) / BN);
            const long waste = ((a.Npad + BN - 1) / BN) * BN - a.Npad;
            const long score = (blocks >= 512 ? 1000000 : blocks * 1000) + BM * BN / 64 - waste * 50;
            if (score > best) { best = score; idx = i; }
        }
    }
    const SplitCfg& c = g_pws_cfgs[idx];
    const int BM = 32 * c.WM, BN = 32 * c.NT * c.WN;
    dim3 grid(xcd_grid((a.M + BM - 1) / BM) * ((a.Npad + BN - 1) / BN));
    g_last_kernel = g_pws_names[idx];
    int i = 0;
#define X(wm, wn, nt, kc)                                                                                  \
    if (i++ == idx) { hipLaunchKernelGGL((gemm_split_kernel<wm, wn, nt, kc>), grid, dim3(256), 0, s, a); return true; }
    YN_PWS_CONFIGS(X)
#undef X
    return false;
}

bool launch_pw_group(const GemmArgs* a, int n, int cfg, hipStream_t s)
{
    if (n < 1 || n > YN_GROUP_MAX) return false;
    for (int p = 0; p < n; ++p)
        if (!a[p].Wsh || !a[p].Wsl || (a[p].K & 1) || (a[p].in_ld & 1) || (a[p].in_off & 1) || a[p].Npad != a[0].Npad) return false;
    int idx = cfg - N_PW_CFGS - N_PWD_CFGS;
    if (idx < 0 || idx >= N_PWS_CFGS) {                     // untuned: the tile gemm_split would pick for the first (largest) problem
        const int nt32 = a[0].Npad / 32;
        idx = 0;
        long best = -1;
        for (int i = 0; i < N_PWS_CFGS; ++i) {
            const SplitCfg& c = g_pws_cfgs[i];
            if (c.KC != 32) continue;
            const long BM = 32 * c.WM, BN = 32 * c.NT * c.WN;
            if (BN > 32 * nt32 + 31 && BN != 32) continue;
            const long blocks = ((a[0].M + BM - 1) / BM) * ((a[0].Npad + BN - 1) / BN);
            const long waste = ((a[0].Npad + BN - 1) / BN) * BN - a[0].Npad;
            const long score = (blocks >= 512 ? 1000000 : blocks * 1000) + BM * BN / 64 - waste * 50;
            if (score > best) { best = score; idx = i; }
        }
    }
    const SplitCfg& c = g_pws_cfgs[idx];
    const int BM = 32 * c.WM, BN = 32 * c.NT * c.WN;
    Group<GemmArgs> g{};
    unsigned tot = 0;
    for (int p = 0; p < YN_GROUP_MAX; ++p) {
        g.first[p] = tot;
        if (p < n) { g.a[p] = a[p]; tot += xcd_grid((a[p].M + BM - 1) / BM) * ((a[p].Npad + BN - 1) / BN); }
    }
    g.first[YN_GROUP_MAX] = tot;
    static char name[64];
    int i = 0;
#define X(wm, wn, nt, kc)                                                                                  \
    if (i++ == idx) { g_last_kernel = "gemm_split_group_kernel<" #wm "," #wn "," #nt "," #kc ">";           \
                      hipLaunchKernelGGL((gemm_split_group_kernel<wm, wn, nt, kc>), dim3(tot), dim3(256), 0, s, g); return true; }
    YN_PWS_CONFIGS(X)
#undef X
    (void)name;
    return false;
}

// false when the layer's strides do not allow 16-byte A loads
static bool launch_pw_direct(const GemmArgs& a, int idx, hipStream_t s)
{
    if (((a.K | a.in_ld | a.in_off) & 3) != 0 || a.K < 4) return false;
    const DirectCfg& c = g_pwd_cfgs[idx];
    const int BM = 32 * c.WM, BN = 32 * c.NT * c.WN;
    dim3 grid(xcd_grid((a.M + BM - 1) / BM) * ((a.Npad + BN - 1) / BN));
    g_last_kernel = g_pwd_names[idx];
    int i = 0;
#define X(wm, wn, nt, d)                                                                                   \
    if (i++ == idx) { hipLaunchKernelGGL((gemm_direct_kernel<wm, wn, nt, d>), grid, dim3(256), 0, s, a); return true; }
    YN_PWD_CONFIGS(X)
#undef X
    return false;
}

static int find_pw_cfg(int wm, int wn, int nt, int kp, int nb)
{
    for (int i = 0; i < N_PW_CFGS; ++i) {
        const TileCfg& c = g_pw_cfgs[i];
        if (c.WM == wm && c.WN == wn && c.NT == nt && c.KP == kp && c.NBUF == nb) return i;
    }
    return 0;
}

// heuristic used when no tuned choice is supplied
static int choose_pw_cfg(int M, int K, int Npad)
{
    const int nt32 = Npad / 32;
    auto pick_nt = [&](int wn, int maxnt) {
        const int need = (nt32 + wn - 1) / wn;             // n-tiles per wave to cover N in one block column
        if (need <= maxnt) return need;
        for (int nt = maxnt; nt >= 1; --nt) if (nt32 % (nt * wn) == 0) return nt;
        return maxnt;
    };
    auto blocks = [&](int wm, int wn, int nt) {
        const long BM = 32 * wm, BN = 32 * nt * wn;
        return ((M + BM - 1) / BM) * ((Npad + BN - 1) / BN);
    };
    int wm = 4, wn = 1, nt = pick_nt(1, 4);
    if (blocks(wm, wn, nt) < 384) {
        const int nt2 = pick_nt(2, 2);
        if (blocks(2, 2, nt2) > blocks(wm, wn, nt)) { wm = 2; wn = 2; nt = nt2; }
        if (blocks(wm, wn, nt) < 256) {
            const int nt3 = pick_nt(4, 2);
            if (blocks(1, 4, nt3) > blocks(wm, wn, nt)) { wm = 1; wn = 4; nt = nt3; }
            if (blocks(wm, wn, nt) < 256 && nt > 1 && blocks(wm, wn, 1) > blocks(wm, wn, nt)) nt = 1;
        }
    }
    const bool one_shot = K <= 64;                          // whole K in one 64-deep chunk, single buffer
    int idx = find_pw_cfg(wm, wn, nt, one_shot ? 32 : 16, one_shot ? 1 : 2);
    const TileCfg& c = g_pw_cfgs[idx];
    if (c.WM != wm || c.WN != wn || c.NT != nt) idx = find_pw_cfg(wm, wn, nt, 16, 2);
    return idx;
}

void launch_pw(const GemmArgs& a, hipStream_t s)
{
    int idx = a.cfg;
    // the persistent form: only when asked for by index (the autotuner times it beside the gemm_split_kernel configurations); not applicable ->
    // the heuristic split configuration
    if (idx == N_PW_CFGS + N_PWD_CFGS + N_PWS_CFGS) { if (launch_pw_pipe(a, s)) return; idx = -1; }
    // layers that carry split packs run on the split-f16 family unless an f32 configuration is requested explicitly
    if (a.Wsh && (idx < 0 || idx >= N_PW_CFGS + N_PWD_CFGS) && launch_pw_split(a, idx < 0 ? -1 : idx - N_PW_CFGS - N_PWD_CFGS, s)) return;
    if (idx >= N_PW_CFGS + N_PWD_CFGS) idx = -1;
    if (idx >= N_PW_CFGS && idx < N_PW_CFGS + N_PWD_CFGS && launch_pw_direct(a, idx - N_PW_CFGS, s)) return;
    if (idx < 0 || idx >= N_PW_CFGS) idx = choose_pw_cfg(a.M, a.K, a.Npad);
    const TileCfg& c = g_pw_cfgs[idx];
    const int BM = 32 * c.WM, BN = 32 * c.NT * c.WN;
    dim3 grid(xcd_grid((a.M + BM - 1) / BM) * ((a.Npad + BN - 1) / BN));        // decoded in the kernel (XCD-aware)
    g_last_kernel = g_pw_names[idx];
    int i = 0;
#define X(wm, wn, nt, kp, nb)                                                                              \
    if (i++ == idx) { hipLaunchKernelGGL((gemm_conv_kernel<wm, wn, nt, 0, kp, nb>), grid, dim3(256), 0, s, a); return; }
    YN_PW_CONFIGS(X)
#undef X
}

// generic (non-LDS-halo) dense 3x3: only two tile shapes are kept
static void launch_gemm_3x3(const GemmArgs& a, hipStream_t s)
{
    const int nt32 = a.Npad / 32;
    if (nt32 % 3 == 0) {
        dim3 grid(xcd_grid((a.M + 127) / 128) * (a.Npad / 96));
        g_last_kernel = "gemm_conv_kernel<4,1,3,1,16,2>";
        hipLaunchKernelGGL((gemm_conv_kernel<4, 1, 3, 1, 16, 2>), grid, dim3(256), 0, s, a);
    } else {
        dim3 grid(xcd_grid((a.M + 127) / 128) * nt32);
        g_last_kernel = "gemm_conv_kernel<4,1,1,1,16,2>";
        hipLaunchKernelGGL((gemm_conv_kernel<4, 1, 1, 1, 16, 2>), grid, dim3(256), 0, s, a);
    }
}

void launch_conv3x3(const GemmArgs& a, hipStream_t s)
{
    // LDS-resident halo kernel when the tile fits (always true for the 96-channel neck up to W ~ 150)
    const int nt32 = a.Npad / 32;
    const int NT = nt32 >= 3 && nt32 % 3 == 0 ? 3 : (nt32 % 2 == 0 ? 2 : 1);
    // the network's 96->96 neck convs: per-tap weight chunks + 16-byte halo staging.  Small maps (few 128-pixel
    // tiles) split N over three block columns instead: the kernel is then latency- not MFMA-bound.
    if (a.Wsh && a.Wsl && a.K == 96 && a.Npad == 96 && a.in_off == 0 && a.in_ld == 96 && (a.N & 3) == 0 && (a.out_ld & 3) == 0 && (a.out_off & 3) == 0 &&
        conv3x3_split_lds(a.W, 1, 1) <= 160 * 1024) {
        static unsigned long long attr_s = 0;
        if (attr_pending(attr_s)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_kernel<3, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_kernel<1, 2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_kernel<1, 1, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_kernel<1, 1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_split_kernel<1, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        const int tiles = (a.M + 127) / 128;
        if (tiles >= 256) {                                // N in one block column; channel halves, two workgroups per CU
            g_last_kernel = "conv3x3_split_kernel<3,2>";
            hipLaunchKernelGGL((conv3x3_split_kernel<3, 2>), dim3(xcd_grid(tiles), 1), dim3(256), conv3x3_split_lds(a.W, 3, 2), s, a);
        } else if (tiles * 3 >= 256) {                     // N over three block columns (more, shorter workgroups).  (Three taps per step here: 38-41 us
            g_last_kernel = "conv3x3_split_kernel<1,2>";    //  against 31 - the 12 KB of weight space cost the third workgroup per CU)
            hipLaunchKernelGGL((conv3x3_split_kernel<1, 2, 1>), dim3(xcd_grid(tiles), 3), dim3(256), conv3x3_split_lds(a.W, 1, 2, 1), s, a);
        } else if (conv3x3_split_lds(a.W, 1, 1, 9) <= 160 * 1024) {   // less than one workgroup per CU: latency-bound, all channels staged at once, nine taps per step
            g_last_kernel = "conv3x3_split_kernel<1,1,9>";
            hipLaunchKernelGGL((conv3x3_split_kernel<1, 1, 9>), dim3(xcd_grid(tiles), 3), dim3(256), conv3x3_split_lds(a.W, 1, 1, 9), s, a);
        } else if (conv3x3_split_lds(a.W, 1, 1, 3) <= 160 * 1024) {   // (wide maps: the halo leaves room for three taps)
            g_last_kernel = "conv3x3_split_kernel<1,1,3>";
            hipLaunchKernelGGL((conv3x3_split_kernel<1, 1, 3>), dim3(xcd_grid(tiles), 3), dim3(256), conv3x3_split_lds(a.W, 1, 1, 3), s, a);
        } else {
            g_last_kernel = "conv3x3_split_kernel<1,1,1>";
            hipLaunchKernelGGL((conv3x3_split_kernel<1, 1, 1>), dim3(xcd_grid(tiles), 3), dim3(256), conv3x3_split_lds(a.W, 1, 1, 1), s, a);
        }
        return;
    }
    if (a.K == 96 && NT == 3 && a.in_off == 0 && a.in_ld == 96 && (a.N & 3) == 0 && (a.out_ld & 3) == 0 && (a.out_off & 3) == 0 &&
        conv3x3_halo_tap_lds(a.W, 96, 3) <= 160 * 1024) {
        static unsigned long long attr_t = 0;
        if (attr_pending(attr_t)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_tap_kernel<3, 96, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_tap_kernel<1, 96, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_tap_kernel<1, 96, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        const int tiles = (a.M + 127) / 128;
        static const int two_per_cu = getenv("YN_C3_SPLIT") ? atoi(getenv("YN_C3_SPLIT")) : 1;
        if (tiles * (a.Npad / 96) >= 256) {
            g_last_kernel = "conv3x3_halo_tap_kernel<3,96,1>";
            hipLaunchKernelGGL((conv3x3_halo_tap_kernel<3, 96, 1>), dim3(xcd_grid(tiles), a.Npad / 96), dim3(256), conv3x3_halo_tap_lds(a.W, 96, 3), s, a);
        } else if (two_per_cu && conv3x3_halo_tap_lds(a.W, 96, 1) > 80 * 1024 && conv3x3_halo_tap_lds(a.W, 96, 1, 2) <= 80 * 1024) {
            // half-tap weight chunks: 6 KB less LDS, which is what lets TWO blocks share a CU on the 26x26 maps (83.6 -> 77.5 KB)
            g_last_kernel = "conv3x3_halo_tap_kernel<1,96,2>";
            hipLaunchKernelGGL((conv3x3_halo_tap_kernel<1, 96, 2>), dim3(xcd_grid(tiles), a.Npad / 32), dim3(256), conv3x3_halo_tap_lds(a.W, 96, 1, 2), s, a);
        } else {
            g_last_kernel = "conv3x3_halo_tap_kernel<1,96,1>";
            hipLaunchKernelGGL((conv3x3_halo_tap_kernel<1, 96, 1>), dim3(xcd_grid(tiles), a.Npad / 32), dim3(256), conv3x3_halo_tap_lds(a.W, 96, 1), s, a);
        }
        return;
    }
    const size_t lds = (a.K % 32 == 0 && a.in_off == 0) ? conv3x3_halo_lds(a.W, a.K, NT) : (size_t)1 << 30;
    if (lds <= 160 * 1024 && a.K / 2 <= 256) {
        static unsigned long long attr = 0;
        if (attr_pending(attr)) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(conv3x3_halo_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        }
        dim3 grid((a.M + 127) / 128, a.Npad / (32 * NT));
        if (NT == 3) { g_last_kernel = "conv3x3_halo_kernel<3>"; hipLaunchKernelGGL(conv3x3_halo_kernel<3>, grid, dim3(256), lds, s, a); }
        else if (NT == 2) { g_last_kernel = "conv3x3_halo_kernel<2>"; hipLaunchKernelGGL(conv3x3_halo_kernel<2>, grid, dim3(256), lds, s, a); }
        else { g_last_kernel = "conv3x3_halo_kernel<1>"; hipLaunchKernelGGL(conv3x3_halo_kernel<1>, grid, dim3(256), lds, s, a); }
        return;
    }
    launch_gemm_3x3(a, s);
}

// -------------------------------------------------------------------------------------------------
// Depthwise 3x3, pad 1, stride 1 or 2.  Thread = (output pixel, channel pair); consecutive threads walk
// the channel pairs of one pixel and then the next pixel, so loads and stores are fully coalesced in
// NHWC.  The nine taps of neighbouring pixels overlap in L1/L2.
// -------------------------------------------------------------------------------------------------
// Depthwise 3x3, one thread = VEC channels x a RUN of R horizontally adjacent output pixels.  The whole input window
// of the run (3 rows x (R+2) or (2R+1) columns) is fetched with unconditional loads at clamped addresses before any of
// it is used — with `if (inside) load` the compiler emits one branch + one vmcnt(0) wait per tap and the kernel sits at
// a third of the HBM rate — and out-of-image taps are zeroed with an opaque bit mask (a select would be sunk back into
// a branch).  VEC = 4 (16-byte accesses) whenever C, the row strides and the channel offsets allow it, else 2.
template <int STRIDE, int VEC, int R>
__device__ __forceinline__ void dwconv3x3_block(const DwArgs& a, unsigned bid, unsigned nblocks)
{
    typedef typename VecT<VEC>::type vec;
    constexpr int NCOL = STRIDE == 1 ? R + 2 : 2 * R + 1;
    const int Ho = (a.H - 1) / STRIDE + 1, Wo = (a.W - 1) / STRIDE + 1;
    const int cv_n = a.C / VEC, segs = (Wo + R - 1) / R;
    const int total = a.B * Ho * segs * cv_n;
    const int i = xcd_block(bid, nblocks) * 256 + threadIdx.x;
    if (i >= total) return;
    const int cv = i % cv_n;
    int q = i / cv_n;
    const int seg = q % segs; q /= segs;
    const int oy = q % Ho, b = q / Ho;
    const int c = cv * VEC, ox0 = seg * R;
    vec col[3][NCOL];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int iy = oy * STRIDE - 1 + ky;
        const bool yok = iy >= 0 && iy < a.H;
        const float* row = a.in + ((size_t)(b * a.H + (yok ? iy : 0)) * a.W) * a.in_ld + a.in_off + c;
#pragma unroll
        for (int j = 0; j < NCOL; ++j) {
            const int ix = ox0 * STRIDE - 1 + j;
            const unsigned mk = opaque_mask(yok && ix >= 0 && ix < a.W);
            col[ky][j] = vmask(*reinterpret_cast<const vec*>(row + (size_t)(ix < 0 ? 0 : (ix >= a.W ? a.W - 1 : ix)) * a.in_ld), mk);
        }
    }
    vec w[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const vec*>(a.w + k * a.C + c);
    const vec bias = *reinterpret_cast<const vec*>(a.bias + c);
    float* orow = a.out + ((size_t)(b * Ho + oy) * Wo) * a.out_ld + a.out_off + c;
#pragma unroll
    for (int o = 0; o < R; ++o) {
        if (ox0 + o >= Wo) break;
        vec acc = bias;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) vfma(acc, col[ky][o * STRIDE + kx], w[ky * 3 + kx]);
        *reinterpret_cast<vec*>(orow + (size_t)(ox0 + o) * a.out_ld) = vact(acc, a.act);
    }
}

template <int STRIDE, int VEC, int R>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(DwArgs a) { dwconv3x3_block<STRIDE, VEC, R>(a, blockIdx.x, gridDim.x); }

template <int R>
__global__ __launch_bounds__(256) void dwconv3x3_group_kernel(Group<DwArgs> g)
{
    unsigned local, nb;
    const int p = group_problem(g.first, blockIdx.x, local, nb);
    dwconv3x3_block<1, 4, R>(g.a[p], local, nb);
}

bool dw_group_ok(const DwArgs* a, int n)
{
    if (n < 1 || n > YN_GROUP_MAX) return false;
    for (int p = 0; p < n; ++p)
        if (a[p].stride != 1 || (a[p].C % 4) || (a[p].in_ld % 4) || (a[p].in_off % 4) || (a[p].out_ld % 4) || (a[p].out_off % 4)) return false;
    return true;
}

void launch_dw_group(const DwArgs* a, int n, hipStream_t s)
{
    auto blocks_for = [&](const DwArgs& q, int r) { return (unsigned)(((long)q.B * q.H * ((q.W + r - 1) / r) * (q.C / 4) + 255) / 256); };
    unsigned all4 = 0;
    for (int p = 0; p < n; ++p) all4 += blocks_for(a[p], 4);
    const int R = all4 >= 1024 ? 4 : 2;                     // run length as launch_dw picks it, for the group as a whole
    Group<DwArgs> g{};
    unsigned tot = 0;
    for (int p = 0; p < YN_GROUP_MAX; ++p) {
        g.first[p] = tot;
        if (p < n) { g.a[p] = a[p]; tot += xcd_grid(blocks_for(a[p], R)); }
    }
    g.first[YN_GROUP_MAX] = tot;
    if (R == 4) { g_last_kernel = "dwconv3x3_group_kernel<4>"; hipLaunchKernelGGL(dwconv3x3_group_kernel<4>, dim3(tot), dim3(256), 0, s, g); }
    else        { g_last_kernel = "dwconv3x3_group_kernel<2>"; hipLaunchKernelGGL(dwconv3x3_group_kernel<2>, dim3(tot), dim3(256), 0, s, g); }
}

void launch_dw(const DwArgs& a, hipStream_t s)
{
    const int Ho = (a.H - 1) / a.stride + 1, Wo = (a.W - 1) / a.stride + 1;
    const bool v4 = (a.C % 4 == 0) && (a.in_ld % 4 == 0) && (a.in_off % 4 == 0) && (a.out_ld % 4 == 0) && (a.out_off % 4 == 0);
    // run length: 4 outputs per thread (8 for the 2-channel variant at stride 1) unless that leaves too few threads to fill the chip
    auto blocks_for = [&](int vec, int r) { return ((long)a.B * Ho * ((Wo + r - 1) / r) * (a.C / vec) + 255) / 256; };
#define YN_DW(ST, V, R) { g_last_kernel = "dwconv3x3_kernel<" #ST "," #V "," #R ">"; \
        hipLaunchKernelGGL((dwconv3x3_kernel<ST, V, R>), dim3(xcd_grid((unsigned)blocks_for(V, R))), dim3(256), 0, s, a); return; }
    if (a.stride == 1) {
        if (v4) { if (blocks_for(4, 4) >= 1024) YN_DW(1, 4, 4) else YN_DW(1, 4, 2) }
        else    { if (blocks_for(2, 8) >= 1024) YN_DW(1, 2, 8) else YN_DW(1, 2, 4) }
    } else {
        if (v4) { if (blocks_for(4, 4) >= 1024) YN_DW(2, 4, 4) else YN_DW(2, 4, 2) }
        else    { if (blocks_for(2, 4) >= 1024) YN_DW(2, 2, 4) else YN_DW(2, 2, 2) }
    }
#undef YN_DW
}

// -------------------------------------------------------------------------------------------------
// Stem: dense 3x3 stride 2 pad 1, Cin = 3, NCHW input -> NHWC output.  One thread per output pixel keeps
// all COUT accumulators in registers; the 27 x COUT weight table is read through the scalar cache.
// -------------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, int B, int H, int W,
                                                    const float* __restrict__ w, const float* __restrict__ bias,
                                                    int act, float* __restrict__ y)
{
    // weights [27][COUT] + bias in LDS (all lanes read the same address => broadcast); a thread owns two
    // horizontally adjacent output pixels so every weight fetched from LDS feeds two FMAs.
    __shared__ __attribute__((aligned(16))) float ws[28 * COUT];
    for (int i = threadIdx.x; i < 27 * COUT; i += 256) ws[i] = w[i];
    for (int i = threadIdx.x; i < COUT; i += 256) ws[27 * COUT + i] = bias[i];
    __syncthreads();
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, Wp = (Wo + 1) / 2;
    const long total = (long)B * Ho * Wp;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int opx = (int)(p % Wp);
    const long q = p / Wp;
    const int oy = (int)(q % Ho);
    const int b = (int)(q / Ho);
    const int ox0 = opx * 2;
    const bool has1 = ox0 + 1 < Wo;
    float acc0[COUT], acc1[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) { acc0[co] = ws[27 * COUT + co]; acc1[co] = acc0[co]; }
    // (ci, ky) is a real loop on purpose: fully unrolled, the compiler hoists all 162 LDS weight reads to the top
    // (648 VGPRs) and spills; per iteration it needs 18 reads / 72 VGPRs.
#pragma unroll 1
    for (int cy = 0; cy < 9; ++cy) {
        const int ci = cy / 3, ky = cy - ci * 3;
        const float* xp = x + ((size_t)b * 3 + ci) * H * W;
        {
            const int iy = oy * 2 - 1 + ky;
            const bool rowok = iy >= 0 && iy < H;
            float in[5];
#pragma unroll
            for (int j = 0; j < 5; ++j) {
                const int ix = ox0 * 2 - 1 + j;
                in[j] = (rowok && ix >= 0 && ix < W) ? xp[(size_t)iy * W + ix] : 0.0f;
            }
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const float4* wr = reinterpret_cast<const float4*>(ws + ((ci * 3 + ky) * 3 + kx) * COUT);
#pragma unroll
                for (int c4 = 0; c4 < COUT / 4; ++c4) {
                    const float4 wv = wr[c4];
                    acc0[c4 * 4 + 0] += in[kx] * wv.x; acc0[c4 * 4 + 1] += in[kx] * wv.y;
                    acc0[c4 * 4 + 2] += in[kx] * wv.z; acc0[c4 * 4 + 3] += in[kx] * wv.w;
                    acc1[c4 * 4 + 0] += in[kx + 2] * wv.x; acc1[c4 * 4 + 1] += in[kx + 2] * wv.y;
                    acc1[c4 * 4 + 2] += in[kx + 2] * wv.z; acc1[c4 * 4 + 3] += in[kx + 2] * wv.w;
                }
            }
        }
    }
    float* yo = y + (((size_t)b * Ho + oy) * Wo + ox0) * COUT;
#pragma unroll
    for (int co = 0; co < COUT; co += 4)
        *reinterpret_cast<float4*>(yo + co) = make_float4(apply_act(acc0[co], act), apply_act(acc0[co + 1], act),
                                                          apply_act(acc0[co + 2], act), apply_act(acc0[co + 3], act));
    if (has1) {
#pragma unroll
        for (int co = 0; co < COUT; co += 4)
            *reinterpret_cast<float4*>(yo + COUT + co) = make_float4(apply_act(acc1[co], act), apply_act(acc1[co + 1], act),
                                                                     apply_act(acc1[co + 2], act), apply_act(acc1[co + 3], act));
    }
}

void launch_stem(const float* x, int B, int H, int W, const float* w, const float* bias, int Cout, int act, float* y, hipStream_t s)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * ((Wo + 1) / 2);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    g_last_kernel = "stem_kernel<24>";
    if (Cout == 24) hipLaunchKernelGGL(stem_kernel<24>, dim3(blocks), dim3(256), 0, s, x, B, H, W, w, bias, act, y);
}

// -------------------------------------------------------------------------------------------------
// Stem conv (3x3 s2, 3->COUT, +bias+act) fused with the 3x3 s2 max pool that follows it
// (backbone/shufflenetv2.py:109-116, 159).  A block owns 8x7 pooled pixels: it computes the 17x15 conv
// pixels they need (255 of 256 threads busy, one conv pixel x COUT channels each) into LDS, then pools from
// LDS.  The 24 x (S/2)^2 conv activation (133 MB per bs=32 step at 416) never goes to memory.
// -------------------------------------------------------------------------------------------------
template <int COUT>
__global__ __launch_bounds__(256) void stem_pool_kernel(const float* __restrict__ x, int B, int H, int W,
                                                         const float* __restrict__ w, const float* __restrict__ bias,
                                                         int act, float* __restrict__ y)
{
    constexpr int PR = 8, PC = 7, CR = 2 * PR + 1, CC = 2 * PC + 1;     // pooled tile, conv tile
    // The weights are wave-uniform: they come through the scalar cache into SGPR operands of the packed FMAs (as broadcast
    // 16-byte LDS reads they took 8 LDS cycles per 4 VALU cycles — the kernel ran at the LDS rate, 58 us); LDS only holds
    // the conv tile the pooling reads.
    __shared__ __attribute__((aligned(16))) float ct[CR * CC * COUT];
    const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1;               // conv output extent
    const int Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;             // pooled extent
    const int tiles_x = (Wp + PC - 1) / PC, tiles_y = (Hp + PR - 1) / PR;
    const int vb = (int)xcd_block(blockIdx.x, gridDim.x);               // XCD-contiguous tile order (yn_internal.h)
    if (vb >= tiles_x * tiles_y * B) return;                             // grid is rounded up to a multiple of 8
    const int tile = vb % (tiles_x * tiles_y), b = vb / (tiles_x * tiles_y);
    const int py0 = (tile / tiles_x) * PR, px0 = (tile % tiles_x) * PC;
    const int t = threadIdx.x;
    const int r = t / CC, c = t - r * CC;
    const int cy = 2 * py0 - 1 + r, cx = 2 * px0 - 1 + c;               // conv pixel of this thread
    const bool live = t < CR * CC && cy >= 0 && cy < Hc && cx >= 0 && cx < Wc;
    // all 27 input values of the thread's conv pixel in ONE batch of unconditional (clamped, masked) loads
    float in[27];
    {
        const int cyc = live ? cy : 0, cxc = live ? cx : 0;
#pragma unroll
        for (int q = 0; q < 9; ++q) {
            const int ci = q / 3, ky = q - ci * 3;
            const float* xp = x + ((size_t)b * 3 + ci) * H * W;
            const int iy = cyc * 2 - 1 + ky;
            const bool rowok = live && iy >= 0 && iy < H;
            const float* xr = xp + (size_t)(iy < 0 ? 0 : (iy >= H ? H - 1 : iy)) * W;
#pragma unroll
            for (int k = 0; k < 3; ++k) {
                const int ix = cxc * 2 - 1 + k;
                const unsigned mk = opaque_mask(rowok && ix >= 0 && ix < W);
                in[q * 3 + k] = __uint_as_float(__float_as_uint(xr[ix < 0 ? 0 : (ix >= W ? W - 1 : ix)]) & mk);
            }
        }
    }
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = bias[co];
#pragma unroll
    for (int i = 0; i < 27; ++i) {
#pragma unroll
        for (int co = 0; co < COUT; ++co) acc[co] += in[i] * w[i * COUT + co];
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = live ? apply_act(acc[co], act) : -INFINITY;   // -inf = max-pool padding
    if (t < CR * CC) {
#pragma unroll
        for (int co = 0; co < COUT; co += 4)
            *reinterpret_cast<float4*>(ct + t * COUT + co) = make_float4(acc[co], acc[co + 1], acc[co + 2], acc[co + 3]);
    }
    __syncthreads();
    constexpr int C4 = COUT / 4;
    for (int i = t; i < PR * PC * C4; i += 256) {
        const int c4 = i % C4, pp = i / C4;
        const int pr = pp / PC, pc = pp - pr * PC;
        const int py = py0 + pr, px = px0 + pc;
        if (py >= Hp || px >= Wp) continue;
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float4 v = *reinterpret_cast<const float4*>(ct + ((2 * pr + dy) * CC + 2 * pc + dx) * COUT + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        *reinterpret_cast<float4*>(y + (((size_t)b * Hp + py) * Wp + px) * COUT + c4 * 4) = m;
    }
}

void launch_stem_pool(const float* x, int B, int H, int W, const float* w, const float* bias, int Cout, int act, float* y, hipStream_t s)
{
    const int Hc = (H - 1) / 2 + 1, Wc = (W - 1) / 2 + 1, Hp = (Hc - 1) / 2 + 1, Wp = (Wc - 1) / 2 + 1;
    const int tiles = ((Wp + 6) / 7) * ((Hp + 7) / 8);
    g_last_kernel = "stem_pool_kernel<24>";
    if (Cout == 24) hipLaunchKernelGGL(stem_pool_kernel<24>, dim3(xcd_grid((unsigned)(tiles * B))), dim3(256), 0, s, x, B, H, W, w, bias, act, y);
}

// 3x3 stride-2 pad-1 max pool (implicit -inf padding), NHWC, thread = (output pixel, 4 channels).
__global__ __launch_bounds__(256) void maxpool_kernel(const float* __restrict__ x, int B, int H, int W, int C, float* __restrict__ y)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int c4n = C >> 2;
    const long total = (long)B * Ho * Wo * c4n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c4 = (int)(i % c4n);
        const long p = i / c4n;
        const int ox = (int)(p % Wo);
        const long q = p / Wo;
        const int oy = (int)(q % Ho);
        const int b = (int)(q / Ho);
        float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                const float4 v = *reinterpret_cast<const float4*>(x + ((size_t)(b * H + iy) * W + ix) * C + c4 * 4);
                m.x = fmaxf(m.x, v.x); m.y = fmaxf(m.y, v.y); m.z = fmaxf(m.z, v.z); m.w = fmaxf(m.w, v.w);
            }
        }
        *reinterpret_cast<float4*>(y + (size_t)p * C + c4 * 4) = m;
    }
}

void launch_maxpool(const float* x, int B, int H, int W, int C, float* y, hipStream_t s)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * Wo * (C >> 2);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    g_last_kernel = "maxpool_kernel";
    hipLaunchKernelGGL(maxpool_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, B, H, W, C, y);
}

// Layout converters (host-shim / test helpers; not on the inference path except nothing).
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, int B, int C, int HW, float* __restrict__ y)
{
    __shared__ float tile[32][33];
    const int b = blockIdx.z;
    const int p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 256 threads: 8 rows per pass
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, p = p0 + tx;
        tile[r][tx] = (c < C && p < HW) ? x[((size_t)b * C + c) * HW + p] : 0.0f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int p = p0 + r, c = c0 + tx;
        if (p < HW && c < C) y[((size_t)b * HW + p) * C + c] = tile[tx][r];
    }
}

void launch_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float* y, hipStream_t s)
{
    const int HW = H * W;
    dim3 grid((HW + 31) / 32, (C + 31) / 32, B);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, s, x, B, C, HW, y);
}

void launch_nhwc_to_nchw(const float* x, int B, int C, int H, int W, float* y, hipStream_t s)
{
    // NHWC [B,HW,C] -> NCHW [B,C,HW] is the same transpose with the roles of (C, HW) swapped
    const int HW = H * W;
    dim3 grid((C + 31) / 32, (HW + 31) / 32, B);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, grid, dim3(256), 0, s, x, B, HW, C, y);
}

// -------------------------------------------------------------------------------------------------
// Weight preparation: BN folding (utils/fuse_conv_bn.py:17-21) + packing, one thread per weight.
// -------------------------------------------------------------------------------------------------
__global__ void fold_pack_kernel(FoldArgs a)
{
    const int per_out = (a.kind == 1) ? a.kk : a.Cin * a.kk;
    const int total = a.Cout * per_out;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < a.Cout) {
        const int co = i;
        float f = 1.0f, bv = a.b ? a.b[co] : 0.0f;
        if (a.gamma) {
            f = a.gamma[co] / sqrtf(a.var[co] + a.eps);
            bv = (bv - a.mean[co]) * f + a.beta[co];
        }
        if (a.b_ref) a.b_ref[co] = bv;
        a.b_packed[co] = bv;
    }
    if (i >= total) return;
    const int co = i / per_out, r = i - co * per_out;
    float f = 1.0f;
    if (a.gamma) f = a.gamma[co] / sqrtf(a.var[co] + a.eps);
    const float v = a.w[i] * f;
    if (a.w_ref) a.w_ref[i] = v;
    if (a.kind == 1) {                          // depthwise: [9][C]
        a.w_packed[r * a.Cout + co] = v;
    } else if (a.kind == 2) {                   // stem: [(ci*3+ky)*3+kx][Cout]
        a.w_packed[r * a.Cout + co] = v;
    } else {                                    // GEMM: k = tap*Cin + ci ; Wp[(k/2)][n][k&1]
        const int ci = r / a.kk, tap = r - ci * a.kk;
        const int k = tap * a.Cin + ci;
        a.w_packed[((size_t)(k >> 1) * a.Npad + co) * 2 + (k & 1)] = v;
        if (a.ws_hi) {                          // the same weight as an exact-to-2^-22 pair of halves: v = hi + lo * 2^-11
            if (a.w_ovf && !(fabsf(v) < 65504.0f)) atomicOr(a.w_ovf, 1u);      // does not fit the split (or is not finite): the caller drops to the f32-MFMA family
            const _Float16 hi = (_Float16)v, lo = (_Float16)((v - (float)hi) * 2048.0f);
            const size_t o = (((size_t)tap * ((a.Cin + 7) >> 3) + (ci >> 3)) * a.Npad + co) * 8 + (ci & 7);
            reinterpret_cast<_Float16*>(a.ws_hi)[o] = hi;
            reinterpret_cast<_Float16*>(a.ws_lo)[o] = lo;
        }
    }
}

void launch_fold_pack(const FoldArgs& a, hipStream_t s)
{
    const int per_out = (a.kind == 1) ? a.kk : a.Cin * a.kk;
    int total = a.Cout * per_out;
    if (total < a.Cout) total = a.Cout;
    hipLaunchKernelGGL(fold_pack_kernel, dim3((total + 255) / 256), dim3(256), 0, s, a);
}

}  // namespace ynk

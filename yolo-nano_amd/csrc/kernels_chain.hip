// kernels_chain.hip — multi-layer tile kernel: unit_chain_kernel (one launch per stride-1 ShuffleV2 unit, cut at the depthwise conv).
#include "yn_internal.h"
#include "yn_device.h"

#include <cstdlib>

namespace ynk {

// -------------------------------------------------------------------------------------------------
// Stride-1 ShuffleV2 units as a chain of ONE kernel per unit (backbone/shufflenetv2.py:53-63, 70-72, 14-28).
//
//   unit i:   x = [x1 | x2],  t = relu(pw1_i(x2)),  y = relu(pw2_i(dw_i(t))),  out = shuffle(cat(x1, y))
//
// Cutting the chain at the unit INPUT needs pw1 on a halo (shuffle_unit_kernel: slower than three kernels).  Cutting it at the
// depthwise conv does not: everything after dw_i is pixel-local, including the next unit's pw1, because the shuffle makes
// x2_{i+1} = interleave(x1[bf/2:], y[bf/2:]) of the SAME pixel.  So the kernel of unit i is
//   1. dw_i on the block's BM pixels, read from global (t_i, neighbours through L1/L2; XCD-contiguous tiles keep them in one
//      L2), each output computed once -> LDS tile T [BM][bf+2];
//   2. GEMM y = relu(T * W2 + b2) (f32 MFMA, A fragments from T, W2 streamed through one LDS buffer with register prefetch);
//   3. y -> T (in place, after a barrier);  interleave pass: out[:, 0:bf] = interleave(x1[0:bf/2], y[0:bf/2]) -> global (the
//      next unit's pass-through half — the only part of `out` a later kernel reads), x2' = interleave(x1[bf/2:], y[bf/2:])
//      -> registers -> T;  (last unit of a stage: the whole `out` row goes to global and the kernel ends here)
//   4. GEMM t_{i+1} = relu(T * W1' + b1') -> global.
// Per unit: 1 launch instead of 3, 4 tensors of [M][bf] through memory instead of 8, no halo recompute.  All sums run in the
// order of the separate kernels (same fma chain in the depthwise conv, same k order in the GEMMs): bit-identical results.
// Measured (tools/phase_timing.sh chain2, stage 3, M = 21 632, bf = 116; cycles per block): depthwise phase 20 k, GEMM 13.1 k each (two
// co-resident blocks share the MFMA pipe: 2 x 7.4 k of MFMA issue), y->T 5.5 k, interleave 3.2 k, epilogue 5.6 k = 39 us per unit
// against 46 us for the three kernels (stage 2: 42 vs 66, stage 4: 47 vs 45).  All blocks of the launch are resident at once and run their phases in lockstep, so the depthwise phase is the
// whole chip fetching its ~35 MB at the same time (bandwidth-bound, MFMAs idle) and the GEMM phases leave the memory system
// idle; overlapping them needs a persistent block that requests tile i+1's window (direct-to-LDS loads) while tile i is in
// its GEMMs — the next step for this kernel.  (Tried: the GEMM weights register-direct from L2 instead of through LDS, which
// removes the chunk barriers — GEMM phase 13.7 k -> 22 k cycles: the 8-byte per-lane weight loads are slower than the barriers;
// and starting the second block of a CU 8 k / 16 k / 24 k cycles late (s_sleep on odd HW wave slots) so that co-resident blocks
// are out of phase: 40.5 / 42.7 / 44.8 us against 42 us, i.e. nothing — with 338 blocks only a third of the CUs hold two.)
// -------------------------------------------------------------------------------------------------
template <int WM, int WN, int NT, int V>
__global__ __launch_bounds__(256, 2) void unit_chain_kernel(ChainArgs a)
{
    typedef typename VecT<V>::type vec;
    constexpr int BM = 32 * WM, BN = 32 * NT * WN, KP = 16, BS = BN * 2, B_PER = KP * BN / 512, G = V / 2;
    constexpr int MAXB = (BM * (BN / (2 * G)) + 255) / 256;     // interleave items per thread and half (upper bound: bf <= BN)
    static_assert(WM * WN == 4 && B_PER >= 1, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float uc_smem[];
    const int bf = a.bf, CS = bf + 2, W = a.W, H = a.H, HW = H * W;
    float* T = uc_smem;                                      // [BM][CS]
    float* Bs = uc_smem + ((BM * CS + 3) & ~3);              // [2][KP][BS]

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    const int m0 = (int)xcd_block(blockIdx.x, gridDim.x) * BM;
    if (m0 >= a.M) return;
    const int nchunks = (bf + 2 * KP - 1) / (2 * KP), kp_total = bf >> 1;

    // GEMM weights: chunks of KP k-pairs through TWO LDS buffers, requested TWO chunks ahead into two register sets, so a
    // chunk's loads have two chunk-times (~4 k cycles) to arrive and there is one barrier per chunk.
    float4 regA[B_PER], regB[B_PER];
    auto prefetch_b = [&](float4 (&reg)[B_PER], const float* Wp, int c) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = t + 256 * i;
            const int kp = idx / (BN / 2), c4 = idx - kp * (BN / 2);
            const int kpg = c * KP + kp;
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (kpg < kp_total) v = *reinterpret_cast<const float4*>(Wp + ((size_t)kpg * a.Npad + c4 * 2) * 2);
            reg[i] = v;
        }
    };
    auto stage_b = [&](float4 (&reg)[B_PER], int buf) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = t + 256 * i;
            const int kp = idx / (BN / 2), c4 = idx - kp * (BN / 2);
            *reinterpret_cast<float4*>(Bs + buf * KP * BS + kp * BS + c4 * 4) = reg[i];
        }
    };
#ifdef YN_EXP_TIMING
    long long TS[8]; int tsn = 0;
#define YN_TS() TS[tsn++] = __builtin_readcyclecounter()
#else
#define YN_TS()
#endif
    YN_TS();
    // the two GEMM biases of this lane's columns: requested now, used tens of microseconds later (a dependent load at the start
    // of an epilogue is a full memory latency with nothing to hide it)
    float bias2[NT], bias1n[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = wn * NT * 32 + nt * 32 + l31;
        bias2[nt] = n < bf ? a.b2[n] : 0.0f;
        bias1n[nt] = (a.Wp1n && n < bf) ? a.b1n[n] : 0.0f;
    }

    // The pass-through half x1 is only needed by the interleave pass after the first GEMM: request it now, so that its
    // latency hides behind the depthwise conv and the GEMM.  Item it = (row, G consecutive j of the first half-row); the same
    // item also owns the j + bf/2 of the second half-row (xg / xl).
    const int hipr = bf / (2 * G), jhi = bf >> 1;          // items per half row; first j of the second half (bf/2 is a multiple of G)
    float2 xg[MAXB], xl[MAXB];                               // G == 1 uses .x only
    auto x1_prefetch = [&]() {
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int it = t + 256 * i;
        const int r = it / hipr, j0 = (it - r * hipr) * G;
        const int m = m0 + r < a.M ? m0 + r : a.M - 1;
        xg[i] = make_float2(0.0f, 0.0f); xl[i] = xg[i];
        if (it < BM * hipr) {
            const float* px = a.x1 + (size_t)m * a.x1_ld + a.x1_off + j0;
            if constexpr (G == 2) {
                xg[i] = *reinterpret_cast<const float2*>(px);
                xl[i] = *reinterpret_cast<const float2*>(px + jhi);
            } else {
                xg[i].x = px[0];
                xl[i].x = px[jhi];
            }
        }
    }
    };

    // ---- 1. depthwise 3x3 of the block's pixels -> T ---------------------------------------------------------
    {
        const int cgn = bf / V, ppl = 256 / cgn;
        const int cg = t % cgn, pl = t / cgn, c = cg * V;
        const bool worker = pl < ppl;                        // 256 is not a multiple of the channel groups: a few threads idle
        {
            // Thread = (channel group, runs of R consecutive flat pixels).  Neighbour (dy, dx) of flat pixel p is flat pixel
            // p + dy*W + dx wherever it exists, so the 3 x (R+2) window of a run is three runs of consecutive pixels: 18 loads
            // (one batch, no masks: addresses clamped into the tensor) feed R = 4 outputs; image borders are handled when the
            // window is used (select 0 per output and tap).
            constexpr int R = 4;
            auto issue = [&](int run, vec (&win)[3][R + 2]) {
                const int q0 = m0 + run * R - 1;
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int i = 0; i < R + 2; ++i) {
                        int q = q0 + (dy - 1) * W + i;
                        q = q < 0 ? 0 : (q >= a.M ? a.M - 1 : q);
                        win[dy][i] = *reinterpret_cast<const vec*>(a.t1 + (size_t)q * a.t1_ld + a.t1_off + c);
                    }
            };
            // the nine taps and the bias of this thread's channel group live in registers (the kernel runs at two waves per SIMD
            // either way; from LDS they cost 72 ds_read_b128 per thread)
            vec w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const vec*>(a.wdw + k * bf + c);
            const vec bias = *reinterpret_cast<const vec*>(a.bdw + c);
            auto finish = [&](int run, vec (&win)[3][R + 2]) {
                const int mrun = m0 + run * R;
                const int rem0 = (mrun < a.M ? mrun : m0) % HW;  // one division per run; the R pixels advance by one column
                int y = rem0 / W, x = rem0 - y * W;
#pragma unroll
                for (int i = 0; i < R; ++i) {
                    const int r = run * R + i;
                    const bool live = mrun + i < a.M;
                    const bool yk[3] = {live && y >= 1, live, live && y + 1 < H};
                    const bool xk[3] = {x >= 1, true, x + 1 < W};
                    vec acc = bias;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const bool ok = yk[ky] && xk[kx];
                            vec v = win[ky][i + kx];
                            if constexpr (V == 4) v = make_float4(ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f);
                            else v = make_float2(ok ? v.x : 0.0f, ok ? v.y : 0.0f);
                            vfma(acc, v, w[ky * 3 + kx]);
                        }
                    acc = vact(acc, a.dw_act);
                    float* d = T + r * CS + c;                   // row stride CS*4 bytes is only 8-byte aligned
                    if constexpr (V == 4) {
                        *reinterpret_cast<float2*>(d) = make_float2(acc.x, acc.y);
                        *reinterpret_cast<float2*>(d + 2) = make_float2(acc.z, acc.w);
                    } else {
                        *reinterpret_cast<float2*>(d) = acc;
                    }
                    if (++x == W) { x = 0; if (++y == H) y = 0; }
                }
            };
            // One window in flight per thread: the latency is hidden by the other blocks of the CU (a second window in flight
            // costs 72 registers and an occupancy step: measured 2x slower).
            vec win[3][R + 2];
            // request order = completion order (vmcnt is in-order): the first window, then the GEMM's first weight chunk, then x1
            if (worker) issue(pl, win);
            prefetch_b(regA, a.Wp2, 0);
            x1_prefetch();
            stage_b(regA, 0);                                // GEMM entry state: chunk 0 in buffer 0, chunk 1 requested into regA
            if (nchunks > 1) prefetch_b(regA, a.Wp2, 1);
            if (worker) {
                for (int run = pl; run < BM / R; run += ppl) {
                    finish(run, win);
                    if (run + ppl < BM / R) issue(run + ppl, win);
                }
            }
        }
        for (int r = t; r < BM; r += 256) *reinterpret_cast<float2*>(T + r * CS + bf) = make_float2(0.0f, 0.0f);   // pad columns (K tail)
    }
    __syncthreads();
    YN_TS();

    // ---- GEMM: acc = T[BM][bf] * Wp (chunk 0 already staged in Bs) --------------------------------------------
    f32x16 acc[NT];
    auto mfma_chunk = [&](int c, int buf) {
        const int krem = bf - c * 2 * KP;
        const int nq = krem >= 2 * KP ? KP / 2 : ((krem + 3) >> 2);
        const float* Ab = T + (wm * 32 + l31) * CS + c * 2 * KP + 2 * h;
        const float* Bb = Bs + buf * KP * BS + (wn * NT * 32 + l31) * 2 + h * BS;
        float2 av = *reinterpret_cast<const float2*>(Ab);
        float2 bv[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const float2*>(Bb + nt * 64);
#pragma unroll
        for (int q = 0; q < KP / 2; ++q) {
            if (q < nq) {                                   // wave-uniform
                float2 av_n = av, bv_n[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv_n[nt] = bv[nt];
                if (q + 1 < nq) {
                    av_n = *reinterpret_cast<const float2*>(Ab + 4 * (q + 1));
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
    };
    // entry state: chunk 0 staged in buffer 0 (visible), chunk 1 requested into regA
    auto gemm = [&](const float* Wp) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][k] = 0.0f;
        for (int c = 0; c < nchunks; c += 2) {
            if (c + 2 < nchunks) prefetch_b(regB, Wp, c + 2);
            mfma_chunk(c, 0);
            if (c + 1 >= nchunks) break;
            stage_b(regA, 1);                               // buffer 1 was last read two chunks ago, before the previous barrier
            __syncthreads();
            if (c + 3 < nchunks) prefetch_b(regA, Wp, c + 3);
            mfma_chunk(c + 1, 1);
            if (c + 2 < nchunks) {
                stage_b(regB, 0);
                __syncthreads();
            }
        }
    };
    gemm(a.Wp2);
    YN_TS();
    if (a.Wp1n) prefetch_b(regA, a.Wp1n, 0);
    __syncthreads();                                        // all waves are done reading T and Bs

    // ---- 3. y = act(acc + b2) -> T (in place) -----------------------------------------------------------------
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = wn * NT * 32 + nt * 32 + l31;
        if (n < bf) {
            const float bias = bias2[nt];
#pragma unroll
            for (int r = 0; r < 16; ++r) T[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * CS + n] = apply_act(acc[nt][r] + bias, a.act2);
        }
    }
    if (a.Wp1n) {
        stage_b(regA, 0);
        if (nchunks > 1) prefetch_b(regA, a.Wp1n, 1);
    }
    __syncthreads();
    YN_TS();

    // interleave pass.  Item = (row, G consecutive j): (x1[j], y[j], ...) -> 2G consecutive output channels 2j..
    // First half-row (j < bf/2) -> global; second half-row -> global too (last unit: whole `out` row) or -> x2' in T.
    float2 yl[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int it = t + 256 * i;
        const int r = it / hipr, j0 = (it - r * hipr) * G, m = m0 + r;
        yl[i] = make_float2(0.0f, 0.0f);
        if (it < BM * hipr) {
            float2 y = make_float2(0.0f, 0.0f);
            if constexpr (G == 2) { y = *reinterpret_cast<const float2*>(T + r * CS + j0); yl[i] = *reinterpret_cast<const float2*>(T + r * CS + jhi + j0); }
            else { y.x = T[r * CS + j0]; yl[i].x = T[r * CS + jhi + j0]; }
            if (m < a.M) {
                float* o = a.out + (size_t)m * a.out_ld + 2 * j0;
                if constexpr (G == 2) {
                    *reinterpret_cast<float4*>(o) = make_float4(xg[i].x, y.x, xg[i].y, y.y);
                    if (!a.Wp1n) *reinterpret_cast<float4*>(o + 2 * jhi) = make_float4(xl[i].x, yl[i].x, xl[i].y, yl[i].y);
                } else {
                    *reinterpret_cast<float2*>(o) = make_float2(xg[i].x, y.x);
                    if (!a.Wp1n) *reinterpret_cast<float2*>(o + 2 * jhi) = make_float2(xl[i].x, yl[i].x);
                }
            }
        }
    }
    if (!a.Wp1n) return;
    __syncthreads();                                        // every read of y is done: T may be overwritten
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {                        // ... and x2' = interleave(x1[bf/2:], y[bf/2:]) into T
        const int it = t + 256 * i;
        const int r = it / hipr, c0 = 2 * (it - r * hipr) * G;
        if (it < BM * hipr) {
            *reinterpret_cast<float2*>(T + r * CS + c0) = make_float2(xl[i].x, yl[i].x);
            if constexpr (G == 2) *reinterpret_cast<float2*>(T + r * CS + c0 + 2) = make_float2(xl[i].y, yl[i].y);
        }
    }
    __syncthreads();

    YN_TS();
    // ---- 4. the next unit's pw1 on x2' -> global ------------------------------------------------------------------
    gemm(a.Wp1n);
    YN_TS();
    GemmArgs e{};
    e.out = a.t1n; e.out_ld = bf; e.out_off = 0; e.M = a.M; e.N = bf; e.Npad = a.Npad; e.bias = a.b1n; e.act = a.act1n; e.pass = nullptr;
    gemm_epilogue<NT>(e, acc, m0 + wm * 32, wn * NT * 32, (bf & 3) == 0, lane, bias1n);
#ifdef YN_EXP_TIMING
    YN_TS();
    if (t == 0 && (blockIdx.x % 97) == 5)
        printf("chain bf %d blk %d start %lld dw %lld gemm1 %lld y %lld interleave %lld gemm2 %lld epi %lld\n", bf, (int)blockIdx.x, TS[0], TS[1] - TS[0], TS[2] - TS[1], TS[3] - TS[2], TS[4] - TS[3], TS[5] - TS[4], TS[6] - TS[5]);
#endif
#undef YN_TS
}

// -------------------------------------------------------------------------------------------------
// The same chain on the split-f16 family (gemm_split_kernel's numerics: x = hi + lo*2^-11, three f16 MFMAs per product, K in chunks
// of 32 in the same order => bit-identical to the three-kernel path on gemm_split_kernel).  With the GEMM phases 4-5x shorter the
// unit is the depthwise (memory) phase plus two short matrix phases, so ONE launch per unit beats three again.  LDS: the depthwise
// output / x2' as two planes of halves P[2][BM][PS] (A operands), the fp32 tile T32 [BM][bf+2] for y (the interleave pass needs y
// in full precision), one buffer of pre-split weight chunks.
// -------------------------------------------------------------------------------------------------
typedef _Float16 uch16;
typedef _Float16 uch16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 uch16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 uch16x2 __attribute__((ext_vector_type(2)));

template <int WM, int WN, int NT, int V>
__global__ __launch_bounds__(256, 2) void unit_chain_split_kernel(ChainArgs a)
{
    typedef typename VecT<V>::type vec;
    constexpr int BM = 32 * WM, BN = 32 * NT * WN, KC = 32, G = V / 2;
    constexpr int B_PER = (2 * 4 * BN + 255) / 256;
    constexpr int MAXB = (BM * (BN / (2 * G)) + 255) / 256;
    static_assert(WM * WN == 4, "4 waves");
    extern __shared__ __attribute__((aligned(16))) float ucs_smem[];
    const int bf = a.bf, CS = bf + 2, W = a.W, H = a.H, HW = H * W;
    const int KQ = (bf + 7) >> 3, PS = plane_stride(bf), nchunks = (bf + KC - 1) / KC;
    float* T32 = ucs_smem;                                              // [BM][CS]
    uch16* Ph = reinterpret_cast<uch16*>(ucs_smem + ((BM * CS + 3) & ~3));  // [BM][PS]
    uch16* Pl = Ph + BM * PS;
    uch16* Bh = Pl + BM * PS;                                           // [4][BN][8], then the lo plane

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    const int m0 = (int)xcd_block(blockIdx.x, gridDim.x) * BM;
    if (m0 >= a.M) return;

#ifdef YN_EXP_TIMING
    long long TS[8]; int tsn = 0;
#define YN_TS() TS[tsn++] = __builtin_readcyclecounter()
#else
#define YN_TS()
#endif
    YN_TS();
    uch16x8 b_reg[B_PER];
    auto prefetch_b = [&](const void* Wh, const void* Wl, int c) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int g = t + 256 * i;
            const int pl = g / (4 * BN), r = g - pl * (4 * BN);
            const int o = r / BN, n = r - o * BN;
            const int kq = c * (KC / 8) + o;
            const bool ok = g < 2 * 4 * BN && kq < KQ && n < a.Npad;
            uch16x8 v = *reinterpret_cast<const uch16x8*>(reinterpret_cast<const uch16*>(pl ? Wl : Wh) + ((size_t)(ok ? kq : 0) * a.Npad + (ok ? n : 0)) * 8);
            if (!ok) {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (uch16)0.0f;
            }
            b_reg[i] = v;
        }
    };
    auto stage_b = [&]() {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int g = t + 256 * i;
            if (g < 2 * 4 * BN) *reinterpret_cast<uch16x8*>(Bh + (size_t)g * 8) = b_reg[i];
        }
    };
    float amax = 0.0f;                                                   // range guard (yn_device.h): largest |value| this thread has split
    auto split_store = [&](int r, int c, float v0, float v1) {           // two adjacent channels of row r -> both planes
        uch16x2 hi, lo;
        amax = range_track(range_track(amax, v0), v1);
        hi[0] = (uch16)v0; hi[1] = (uch16)v1;
        lo[0] = (uch16)((v0 - (float)hi[0]) * 2048.0f); lo[1] = (uch16)((v1 - (float)hi[1]) * 2048.0f);
        *reinterpret_cast<uch16x2*>(Ph + r * PS + c) = hi;
        *reinterpret_cast<uch16x2*>(Pl + r * PS + c) = lo;
    };

    float bias2[NT], bias1n[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = wn * NT * 32 + nt * 32 + l31;
        bias2[nt] = n < bf ? a.b2[n] : 0.0f;
        bias1n[nt] = (a.Wp1n && n < bf) ? a.b1n[n] : 0.0f;
    }
    const int hipr = bf / (2 * G), jhi = bf >> 1;
    float2 xg[MAXB], xl[MAXB];
    auto x1_prefetch = [&]() {
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int it = t + 256 * i;
        const int r = it / hipr, j0 = (it - r * hipr) * G;
        const int m = m0 + r < a.M ? m0 + r : a.M - 1;
        xg[i] = make_float2(0.0f, 0.0f); xl[i] = xg[i];
        if (it < BM * hipr) {
            const float* px = a.x1 + (size_t)m * a.x1_ld + a.x1_off + j0;
            if constexpr (G == 2) {
                xg[i] = *reinterpret_cast<const float2*>(px);
                xl[i] = *reinterpret_cast<const float2*>(px + jhi);
            } else {
                xg[i].x = px[0];
                xl[i].x = px[jhi];
            }
        }
    }
    };

    // ---- 1. depthwise 3x3 of the block's pixels -> split planes (the same fma chain as dwconv3x3_kernel) ------------------------
    {
        const int cgn = bf / V, ppl = 256 / cgn;
        const int cg = t % cgn, pl = t / cgn, c = cg * V;
        const bool worker = pl < ppl;
        constexpr int R = 4;
        auto issue = [&](int run, vec (&win)[3][R + 2]) {
            const int q0 = m0 + run * R - 1;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int i = 0; i < R + 2; ++i) {
                    int q = q0 + (dy - 1) * W + i;
                    q = q < 0 ? 0 : (q >= a.M ? a.M - 1 : q);
                    win[dy][i] = *reinterpret_cast<const vec*>(a.t1 + (size_t)q * a.t1_ld + a.t1_off + c);
                }
        };
        vec w[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const vec*>(a.wdw + k * bf + c);
        const vec bias = *reinterpret_cast<const vec*>(a.bdw + c);
        auto finish = [&](int run, vec (&win)[3][R + 2]) {
            const int mrun = m0 + run * R;
            const int rem0 = (mrun < a.M ? mrun : m0) % HW;
            int y = rem0 / W, x = rem0 - y * W;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int r = run * R + i;
                const bool live = mrun + i < a.M;
                const bool yk[3] = {live && y >= 1, live, live && y + 1 < H};
                const bool xk[3] = {x >= 1, true, x + 1 < W};
                vec acc = bias;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const bool ok = yk[ky] && xk[kx];
                        vec v = win[ky][i + kx];
                        if constexpr (V == 4) v = make_float4(ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f);
                        else v = make_float2(ok ? v.x : 0.0f, ok ? v.y : 0.0f);
                        vfma(acc, v, w[ky * 3 + kx]);
                    }
                acc = vact(acc, a.dw_act);
                if constexpr (V == 4) { split_store(r, c, acc.x, acc.y); split_store(r, c + 2, acc.z, acc.w); }
                else split_store(r, c, acc.x, acc.y);
                if (++x == W) { x = 0; if (++y == H) y = 0; }
            }
        };
        vec win[3][R + 2];
        if (worker) issue(pl, win);
        prefetch_b(a.Ws2h, a.Ws2l, 0);
        x1_prefetch();
        stage_b();
#ifdef YN_EXP_TIMING
        long long td[6]; int tdn = 0; td[tdn++] = __builtin_readcyclecounter();
#endif
        if (nchunks > 1) prefetch_b(a.Ws2h, a.Ws2l, 1);
        if (worker) {
            for (int run = pl; run < BM / R; run += ppl) {
                finish(run, win);
#ifdef YN_EXP_TIMING
                if (tdn < 5) td[tdn++] = __builtin_readcyclecounter();
#endif
                if (run + ppl < BM / R) issue(run + ppl, win);
            }
        }
#ifdef YN_EXP_TIMING
        if (t == 0 && (blockIdx.x % 97) == 5) printf("chaindw bf %d blk %d start->stage_b %lld run0 %lld run1 %lld\n", bf, (int)blockIdx.x, td[0] - TS[0], td[1] - td[0], tdn > 2 ? td[2] - td[1] : 0LL);
#endif
        // K tail: the columns [bf, PS) of both planes are zero (they meet zero weight rows, but must not be NaN bit patterns)
        const int padn = PS - bf;
        for (int i = t; i < BM * padn; i += 256) { const int r = i / padn, c2 = bf + i - r * padn; Ph[r * PS + c2] = (uch16)0.0f; Pl[r * PS + c2] = (uch16)0.0f; }
    }
    __syncthreads();
    YN_TS();

    f32x16 acc0[NT], acc1[NT];
    // entry state: chunk 0 staged (visible), chunk 1 requested into b_reg
    auto gemm = [&](const void* Wh, const void* Wl) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) { acc0[i][k] = 0.0f; acc1[i][k] = 0.0f; }
        for (int c = 0; c < nchunks; ++c) {
            const uch16* Ahb = Ph + (wm * 32 + l31) * PS + c * KC + h * 8;
            const uch16* Alb = Pl + (wm * 32 + l31) * PS + c * KC + h * 8;
            const uch16* Bhb = Bh + (size_t)(h * BN + wn * NT * 32 + l31) * 8;
            const uch16* Blb = Bhb + 4 * BN * 8;
#pragma unroll
            for (int ks = 0; ks < KC / 16; ++ks) {
                if (c * (KC / 8) + ks * 2 >= KQ) break;                 // wave-uniform: this 16-deep step lies beyond the (zero-padded) K
                const uch16x8 ah = *reinterpret_cast<const uch16x8*>(Ahb + ks * 16);
                const uch16x8 al = *reinterpret_cast<const uch16x8*>(Alb + ks * 16);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    const uch16x8 bh = *reinterpret_cast<const uch16x8*>(Bhb + (size_t)(ks * 2 * BN + nt * 32) * 8);
                    const uch16x8 bl = *reinterpret_cast<const uch16x8*>(Blb + (size_t)(ks * 2 * BN + nt * 32) * 8);
                    acc0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0[nt], 0, 0, 0);
                    acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[nt], 0, 0, 0);
                    acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[nt], 0, 0, 0);
                }
            }
            if (c + 1 < nchunks) {
                __syncthreads();                                        // every wave is done with this chunk's weights
                stage_b();
                __syncthreads();
                if (c + 2 < nchunks) prefetch_b(Wh, Wl, c + 2);
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[nt][r] = __builtin_fmaf(acc1[nt][r], 1.0f / 2048.0f, acc0[nt][r]);
    };
    gemm(a.Ws2h, a.Ws2l);
    YN_TS();
    if (a.Wp1n) prefetch_b(a.Ws1h, a.Ws1l, 0);
    __syncthreads();                                                    // all waves are done reading the planes and the weights

    // ---- 3. y = act(acc + b2) -> T32 ----------------------------------------------------------------------------------------
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = wn * NT * 32 + nt * 32 + l31;
        if (n < bf) {
            const float bias = bias2[nt];
#pragma unroll
            for (int r = 0; r < 16; ++r) T32[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h) * CS + n] = apply_act(acc0[nt][r] + bias, a.act2);
        }
    }
    if (a.Wp1n) {
        stage_b();
        if (nchunks > 1) prefetch_b(a.Ws1h, a.Ws1l, 1);
    }
    __syncthreads();

    // interleave pass (as unit_chain_kernel): first half-row -> global; second half-row -> global (last unit) or x2' -> planes
    float2 yl[MAXB];
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int it = t + 256 * i;
        const int r = it / hipr, j0 = (it - r * hipr) * G, m = m0 + r;
        yl[i] = make_float2(0.0f, 0.0f);
        if (it < BM * hipr) {
            float2 y = make_float2(0.0f, 0.0f);
            if constexpr (G == 2) { y = *reinterpret_cast<const float2*>(T32 + r * CS + j0); yl[i] = *reinterpret_cast<const float2*>(T32 + r * CS + jhi + j0); }
            else { y.x = T32[r * CS + j0]; yl[i].x = T32[r * CS + jhi + j0]; }
            if (m < a.M) {
                float* o = a.out + (size_t)m * a.out_ld + 2 * j0;
                if constexpr (G == 2) {
                    *reinterpret_cast<float4*>(o) = make_float4(xg[i].x, y.x, xg[i].y, y.y);
                    if (!a.Wp1n) *reinterpret_cast<float4*>(o + 2 * jhi) = make_float4(xl[i].x, yl[i].x, xl[i].y, yl[i].y);
                } else {
                    *reinterpret_cast<float2*>(o) = make_float2(xg[i].x, y.x);
                    if (!a.Wp1n) *reinterpret_cast<float2*>(o + 2 * jhi) = make_float2(xl[i].x, yl[i].x);
                }
            }
        }
    }
    YN_TS();
    if (!a.Wp1n) { range_report(a.ovf, amax); return; }
    // x2' = interleave(x1[bf/2:], y[bf/2:]) -> the planes (free since the first GEMM; their pad columns are still zero)
#pragma unroll
    for (int i = 0; i < MAXB; ++i) {
        const int it = t + 256 * i;
        const int r = it / hipr, c0 = 2 * (it - r * hipr) * G;
        if (it < BM * hipr) {
            split_store(r, c0, xl[i].x, yl[i].x);
            if constexpr (G == 2) split_store(r, c0 + 2, xl[i].y, yl[i].y);
        }
    }
    __syncthreads();

    // ---- 4. the next unit's pw1 on x2' -> global ------------------------------------------------------------------------------
    YN_TS();
    gemm(a.Ws1h, a.Ws1l);
    YN_TS();
    GemmArgs e{};
    e.out = a.t1n; e.out_ld = bf; e.out_off = 0; e.M = a.M; e.N = bf; e.Npad = a.Npad; e.bias = a.b1n; e.act = a.act1n; e.pass = nullptr;
    gemm_epilogue<NT>(e, acc0, m0 + wm * 32, wn * NT * 32, (bf & 3) == 0, lane, bias1n);
    range_report(a.ovf, amax);
#ifdef YN_EXP_TIMING
    YN_TS();
    if (t == 0 && (blockIdx.x % 97) == 5)
        printf("chains bf %d blk %d dw %lld gemm1 %lld y+interleave %lld x2split %lld gemm2 %lld epi %lld total %lld\n", bf, (int)blockIdx.x, TS[1] - TS[0], TS[2] - TS[1],
               TS[3] - TS[2], TS[4] - TS[3], TS[5] - TS[4], TS[6] - TS[5], TS[6] - TS[0]);
#endif
#undef YN_TS
}

// -------------------------------------------------------------------------------------------------
// unit_chain_split_kernel, second form (round 3).  What changed against the kernel above, and why (tools/phase_timing.sh chain2: of a
// block's ~50 k cycles only the first ~14 k move data from HBM; the rest is a chain of latency-bound phases with the memory idle):
//   * the fp32 tile T32 and its two passes are gone.  The accumulator layout already gives every lane ONE column n and 16 rows:
//     the lane loads x1[row][n] itself (16 * NT scalar loads at kernel start, 128-byte rows per half-wavefront), and after the
//     first GEMM turns each accumulator value straight into its final place - out[row][2n .. 2n+1] = (x1, y) as one 8-byte
//     store (columns n < bf/2 - or all of them in the last unit of a stage), or x2'[2(n - bf/2) ..] = split(x1, y) into the
//     operand planes (columns n >= bf/2).  One barrier and ~9 k cycles of LDS round trips less per block; LDS 78 -> 64 KB at bf = 116.
//   * (measured and dropped: a BALANCED launch - 96-row tiles on six wavefronts, ceil(M / CUs) = 88 live pixels each, one workgroup
//     per CU instead of 64-pixel tiles that leave 82 CUs with two: the kernel went 25.7 -> 25.2 us, because six wavefronts sharing the
//     LDS and matrix pipes stretch both GEMM phases from 6 k to 8.2 k cycles, and the 81 KB workgroup cost 6 % of the four-stream
//     throughput, 39.2 k -> 36.8 k images/s)
//   * K goes through LDS in chunks of KC = 64 instead of 32: half the barrier pairs inside the two GEMMs (two chunks at bf = 116;
//     a chunk round costs ~2 k cycles whatever it multiplies).  Same k order (16-deep steps in sequence, zero-padded tail skipped):
//     bit-identical to the first form and to gemm_split_kernel.
// -------------------------------------------------------------------------------------------------
#ifndef YN_UC2_OCC_NARROW
#define YN_UC2_OCC_NARROW 3                                 // workgroups per CU the 58-channel (stage 2) instantiation is compiled for
#endif
template <int WM, int WN, int NT, int V, int D>             // D: 16-deep k-steps of weight fragments in flight per wavefront
__global__ __launch_bounds__(64 * WM * WN, (WM * WN != 4 ? 1 : ((NT == 1 && V == 2) ? YN_UC2_OCC_NARROW : 2))) void unit_chain2_kernel(ChainArgs a)
{
    typedef typename VecT<V>::type vec;
    constexpr int BM = 32 * WM, BN = 32 * NT * WN, NTHR = 64 * WM * WN;
    extern __shared__ __attribute__((aligned(16))) float uc2_smem[];
    const int bf = a.bf, W = a.W, H = a.H, HW = H * W;
    const int KQ = (bf + 7) >> 3, PS = plane_stride(bf), S = (KQ + 1) >> 1;     // S: 16-deep k-steps of a GEMM
    uch16* Ph = reinterpret_cast<uch16*>(uc2_smem);                     // [BM][PS]
    uch16* Pl = Ph + BM * PS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    const int m0 = (int)xcd_block(blockIdx.x, gridDim.x) * BM;
    if (m0 >= a.M) return;
    const int nrows = a.M - m0 < BM ? a.M - m0 : BM;                    // live rows of this tile

#ifdef YN_EXP_TIMING
    long long TS[8]; int tsn = 0;
#define YN_TS() TS[tsn++] = __builtin_readcyclecounter()
#else
#define YN_TS()
#endif
    YN_TS();
    // Round 4: the GEMM weights never pass through LDS (down2_kernel's finding): a lane's B fragment of a k-step is 16 contiguous bytes
    // of the pre-split pack, so every wavefront streams the fragments of ITS columns straight from L2 into registers, D k-steps ahead - no
    // chunk barriers, no staging pass, half the LDS (the first form walked K in chunks of 64 through LDS: two barrier rounds of ~2 k
    // cycles per GEMM whatever they multiplied, and a 2.3 k staging pass between the GEMMs).  Same k order (16-deep steps in sequence):
    // the same bits.
    uch16x8 bq[D][NT][2];
    auto load_b = [&](const void* Wh_, const void* Wl_, int s, uch16x8 (&dst)[NT][2]) {
        const uch16* Wh = reinterpret_cast<const uch16*>(Wh_);
        const uch16* Wl = reinterpret_cast<const uch16*>(Wl_);
        // NO masks (round 5): a masked load is USED where it is issued - eight v_and per fragment - and hipcc then waits for it there: every k-step
        // of the walk drained the vector-memory counter (vmcnt(3) ... vmcnt(0) in front of its MFMAs) and cost a full L2 round trip, D deep or
        // not.  Clamped addresses instead: an octet past K (kq = KQ, odd octet counts) meets the planes' zero K tail, a column past Npad is
        // never stored, a step past the walk reads its A fragment from the zero tail (a_col below).  Finite weights x 0: exact zeros, as before.
        const int kq = min(s * 2 + h, KQ - 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = min(wn * NT * 32 + nt * 32 + l31, a.Npad - 1);
            const size_t off = ((size_t)kq * a.Npad + n) * 8;
            dst[nt][0] = *reinterpret_cast<const uch16x8*>(Wh + off);
            dst[nt][1] = *reinterpret_cast<const uch16x8*>(Wl + off);
        }
    };
    auto prime_b = [&](const void* Wh, const void* Wl) {
#pragma unroll
        for (int j = 0; j < D; ++j) load_b(Wh, Wl, j, bq[j]);
    };
    float amax = 0.0f;                                                   // range guard (yn_device.h)
    auto split_store = [&](int r, int c, float v0, float v1) {           // two adjacent channels of row r -> both planes
        uch16x2 hi, lo;
        amax = range_track(range_track(amax, v0), v1);
        hi[0] = (uch16)v0; hi[1] = (uch16)v1;
        lo[0] = (uch16)((v0 - (float)hi[0]) * 2048.0f); lo[1] = (uch16)((v1 - (float)hi[1]) * 2048.0f);
        *reinterpret_cast<uch16x2*>(Ph + r * PS + c) = hi;
        *reinterpret_cast<uch16x2*>(Pl + r * PS + c) = lo;
    };

    float bias2[NT], bias1n[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = wn * NT * 32 + nt * 32 + l31;
        bias2[nt] = n < bf ? a.b2[n] : 0.0f;
        bias1n[nt] = (a.Wp1n && n < bf) ? a.b1n[n] : 0.0f;
    }
    const int jhi = bf >> 1;
    // the pass-through half in the ACCUMULATOR layout: x1v[nt][r] = x1[row(r)][n(nt)] (needed after the first GEMM)
    float x1v[NT][16];
    // wave-uniform base pointers + 32-bit lane offsets: the loads / stores below then use the saddr form instead of a 64-bit
    // multiply-add per access
    const char* x1_base = reinterpret_cast<const char*>(a.x1 + (size_t)m0 * a.x1_ld + a.x1_off);
    auto x1_prefetch = [&]() {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = wn * NT * 32 + nt * 32 + l31;
            const bool nok = n < bf;
            const unsigned mk = opaque_mask(nok);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int rr = row < nrows ? row : nrows - 1;           // clamped into the tile (idle rows are never stored)
                const unsigned off = (unsigned)(rr * a.x1_ld + (nok ? n : 0)) * 4u;
                x1v[nt][r] = __uint_as_float(__float_as_uint(*reinterpret_cast<const float*>(x1_base + off)) & mk);
            }
        }
    };

    // ---- 1. depthwise 3x3 of the block's pixels -> split planes (the same fma chain as dwconv3x3_kernel) ------------------------
    {
        const int cgn = bf / V, ppl = NTHR / cgn;
        const int cg = t % cgn, pl = t / cgn, c = cg * V;
        const bool worker = pl < ppl;
        constexpr int R = 4;
        auto issue = [&](int run, vec (&win)[3][R + 2]) {
            const int q0 = m0 + run * R - 1;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int i = 0; i < R + 2; ++i) {
                    int q = q0 + (dy - 1) * W + i;
                    q = q < 0 ? 0 : (q >= a.M ? a.M - 1 : q);
                    win[dy][i] = *reinterpret_cast<const vec*>(a.t1 + (size_t)q * a.t1_ld + a.t1_off + c);
                }
        };
        vec w[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const vec*>(a.wdw + k * bf + c);
        const vec bias = *reinterpret_cast<const vec*>(a.bdw + c);
        auto finish = [&](int run, vec (&win)[3][R + 2]) {
            const int mrun = m0 + run * R;
            const int rem0 = (mrun < a.M ? mrun : m0) % HW;
            int y = rem0 / W, x = rem0 - y * W;
#pragma unroll
            for (int i = 0; i < R; ++i) {
                const int r = run * R + i;
                const bool live = r < nrows;
                const bool yk[3] = {live && y >= 1, live, live && y + 1 < H};
                const bool xk[3] = {x >= 1, true, x + 1 < W};
                vec acc = bias;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const bool ok = yk[ky] && xk[kx];
                        vec v = win[ky][i + kx];
                        if constexpr (V == 4) v = make_float4(ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f);
                        else v = make_float2(ok ? v.x : 0.0f, ok ? v.y : 0.0f);
                        vfma(acc, v, w[ky * 3 + kx]);
                    }
                acc = vact(acc, a.dw_act);
                if constexpr (V == 4) { split_store(r, c, acc.x, acc.y); split_store(r, c + 2, acc.z, acc.w); }
                else split_store(r, c, acc.x, acc.y);
                if (++x == W) { x = 0; if (++y == H) y = 0; }
            }
        };
        const int nruns = (nrows + R - 1) / R;
        prime_b(a.Ws2h, a.Ws2l);                            // (requested before the windows: they return first)
        // (both windows of a thread in flight at once - the registers are there since the weights stopped passing through LDS - measured
        //  the same: 24.5 vs 25.1 us per stage-3 unit, 40.8 vs 41.2 k images/s; one window keeps the kernel at 203 registers)
        vec win[3][R + 2];
        if (worker && pl < nruns) issue(pl, win);
        if (worker) {
            for (int run = pl; run < nruns; run += ppl) {
                finish(run, win);
                if (run + ppl < nruns) issue(run + ppl, win);
            }
        }
        // requested only now: the window registers are free again, and these loads (x1 is a quarter of the unit's traffic) fly during
        // the first GEMM, when the memory system would otherwise sit idle
        x1_prefetch();
        // K tail: the columns [bf, PS) of both planes are zero (they meet zero weight rows, but must not be NaN bit patterns)
        const int padn = PS - bf;
        for (int i = t; i < BM * padn; i += NTHR) { const int r = i / padn, c2 = bf + i - r * padn; Ph[r * PS + c2] = (uch16)0.0f; Pl[r * PS + c2] = (uch16)0.0f; }
    }
    __syncthreads();
    YN_TS();

    f32x16 acc0[NT], acc1[NT];
    // entry state: the fragments of steps 0 .. D-1 of this matrix are in flight (prime_b).  Steps past the last one (at most D - 1 of them)
    // multiply the zero K tail of the A rows by whatever fragment the clamped load brought: exact zeros, no branch around the loads
    auto gemm = [&](const void* Wh, const void* Wl) {
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) { acc0[i][k] = 0.0f; acc1[i][k] = 0.0f; }
        for (int s0 = 0; s0 < S; s0 += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int s = s0 + j;
                const int a_col = s < S ? s * 16 + h * 8 : PS - 8;      // past the walk: the last 16 bytes of the row's zero K tail (plane_stride)
                const uch16x8 ah = *reinterpret_cast<const uch16x8*>(Ph + (wm * 32 + l31) * PS + a_col);
                const uch16x8 al = *reinterpret_cast<const uch16x8*>(Pl + (wm * 32 + l31) * PS + a_col);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[j][nt][0], acc0[nt], 0, 0, 0);
                    acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[j][nt][1], acc1[nt], 0, 0, 0);
                    acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bq[j][nt][0], acc1[nt], 0, 0, 0);
                }
                load_b(Wh, Wl, s + D, bq[j]);                   // (clamped beyond the last step)
                __builtin_amdgcn_sched_barrier(0);              // HERE, behind the MFMAs that freed the slot: hipcc sank all D steps' loads to the end of
            }                                                   // the unrolled group, where the next group needs them at once (a round trip per group)
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[nt][r] = __builtin_fmaf(acc1[nt][r], 1.0f / 2048.0f, acc0[nt][r]);
    };
    gemm(a.Ws2h, a.Ws2l);
    YN_TS();
    if (a.Wp1n) prime_b(a.Ws1h, a.Ws1l);                               // the next GEMM's first fragments fly during the epilogue
    __syncthreads();                                                    // all waves are done reading the planes
#ifdef YN_EXP_TIMING
    const long long t_bar = __builtin_readcyclecounter();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t_x1 = __builtin_readcyclecounter();
#endif

    // ---- 2. y = act(acc + b2) straight to its final place: (x1, y) pairs -> global, or split into the planes as x2' ---------------
    const bool last = a.Wp1n == nullptr;
    const bool uniform8 = (nrows & 7) == 0;                             // wave-uniform: a row group of 8 is live or idle as a whole
    char* out_base = reinterpret_cast<char*>(a.out + (size_t)m0 * a.out_ld);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = wn * NT * 32 + nt * 32 + l31;
        const float bias = bias2[nt];
        const bool to_global = n < (last ? bf : jhi);
        const bool to_plane = !last && n >= jhi && n < bf;
        float y[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) y[r] = apply_act(acc0[nt][r] + bias, a.act2);
        // ONE exec region per destination, scalar branches per row group (a per-value `if` compiles to a branch + waits around every
        // store: measured 200 cycles each)
        if (uniform8) {
#pragma unroll
            for (int g8 = 0; g8 < 4; ++g8) {
                if (wm * 32 + 8 * g8 < nrows) {                         // scalar condition
                    if (to_global) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int row = wm * 32 + q + 8 * g8 + 4 * h;
                            *reinterpret_cast<float2*>(out_base + (unsigned)(row * a.out_ld + 2 * n) * 4u) = make_float2(x1v[nt][4 * g8 + q], y[4 * g8 + q]);
                        }
                    }
                    if (to_plane) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) split_store(wm * 32 + q + 8 * g8 + 4 * h, 2 * (n - jhi), x1v[nt][4 * g8 + q], y[4 * g8 + q]);
                    }
                }
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (row < nrows) {
                    if (to_global) *reinterpret_cast<float2*>(out_base + (unsigned)(row * a.out_ld + 2 * n) * 4u) = make_float2(x1v[nt][r], y[r]);
                    if (to_plane) split_store(row, 2 * (n - jhi), x1v[nt][r], y[r]);
                }
            }
        }
    }
    YN_TS();
    if (last) { range_report(a.ovf, amax); return; }
    __syncthreads();

    // ---- 3. the next unit's pw1 on x2' -> global -------------------------------------------------------------------------------
    YN_TS();
    gemm(a.Ws1h, a.Ws1l);
    YN_TS();
    GemmArgs e{};
    e.out = a.t1n; e.out_ld = bf; e.out_off = 0; e.M = m0 + nrows; e.N = bf; e.Npad = a.Npad; e.bias = a.b1n; e.act = a.act1n; e.pass = nullptr;      // rows beyond the tile's live ones are not stored
    gemm_epilogue<NT>(e, acc0, m0 + wm * 32, wn * NT * 32, (bf & 3) == 0, lane, bias1n);
    range_report(a.ovf, amax);
#ifdef YN_EXP_TIMING
    YN_TS();
    if (t == 0 && (blockIdx.x % 97) == 5)
        printf("chain2 bf %d blk %d dw %lld gemm1 %lld fused-epi %lld (barrier %lld x1wait %lld work %lld) stage %lld gemm2 %lld epi %lld total %lld\n", bf, (int)blockIdx.x, TS[1] - TS[0], TS[2] - TS[1],
               TS[3] - TS[2], t_bar - TS[2], t_x1 - t_bar, TS[3] - t_x1, TS[4] - TS[3], TS[5] - TS[4], TS[6] - TS[5], TS[6] - TS[0]);
#endif
#undef YN_TS
}

static size_t unit_chain2_lds(int bf, int BM)
{
    const int PS = plane_stride(bf);
    return (size_t)2 * BM * PS * 2;                          // the two operand planes; the weights go through registers
}

static size_t unit_chain_split_lds(int bf, int BM, int BN)
{
    const int PS = plane_stride(bf);
    return (size_t)((BM * (bf + 2) + 3) & ~3) * sizeof(float) + ((size_t)2 * BM * PS + (size_t)2 * 4 * BN * 8) * 2;
}

static size_t unit_chain_lds(int bf, int BM, int BN) { return ((size_t)((BM * (bf + 2) + 3) & ~3) + (size_t)2 * 16 * BN * 2) * sizeof(float); }

// false when no instantiated tile covers the shape (Npad must be one block column; bf % 4 == 0 or the 2-channel variant)
// s == nullptr && dry: only answer whether an instantiated tile covers the shape (run_unit_chain asks before it launches anything)
static bool unit_chain_dispatch(const ChainArgs& a, hipStream_t s, bool dry)
{
    if ((a.bf & 1) || a.Npad < a.bf) return false;
    const bool v4 = (a.bf & 3) == 0 && ((a.t1_ld | a.t1_off | a.x1_ld | a.x1_off | a.out_ld) & 3) == 0;
    if (!v4 && (((a.t1_ld | a.t1_off | a.out_ld) & 1) != 0 || 256 / (a.bf / 2) < 1)) return false;
    if (v4 && 256 / (a.bf / 4) < 1) return false;
#define YN_UC(WMv, WNv, NTv, Vv)                                                                                       \
    {                                                                                                                  \
        constexpr int BM = 32 * WMv, BN = 32 * NTv * WNv;                                                              \
        const size_t lds = unit_chain_lds(a.bf, BM, BN);                                                               \
        if (lds > 160 * 1024) return false;                                                                            \
        if (dry) return true;                                                                                          \
        static unsigned long long attr = 0;                                                                                      \
        if (attr_pending(attr)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(unit_chain_kernel<WMv, WNv, NTv, Vv>),    \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); } \
        set_last_kernel_name("unit_chain_kernel<" #WMv "," #WNv "," #NTv "," #Vv ">");                                      \
        hipLaunchKernelGGL((unit_chain_kernel<WMv, WNv, NTv, Vv>), dim3(xcd_grid((a.M + BM - 1) / BM)), dim3(256), lds, s, a); \
        return true;                                                                                                   \
    }
#define YN_UCS(WMv, WNv, NTv, Vv)                                                                                      \
    {                                                                                                                  \
        constexpr int BM = 32 * WMv, BN = 32 * NTv * WNv;                                                              \
        const size_t lds = unit_chain_split_lds(a.bf, BM, BN);                                                         \
        if (lds > 160 * 1024) return false;                                                                            \
        if (dry) return true;                                                                                          \
        static unsigned long long attr = 0;                                                                            \
        if (attr_pending(attr)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(unit_chain_split_kernel<WMv, WNv, NTv, Vv>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }              \
        set_last_kernel_name("unit_chain_split_kernel<" #WMv "," #WNv "," #NTv "," #Vv ">");                            \
        hipLaunchKernelGGL((unit_chain_split_kernel<WMv, WNv, NTv, Vv>), dim3(xcd_grid((a.M + BM - 1) / BM)), dim3(256), lds, s, a); \
        return true;                                                                                                   \
    }
#define YN_UC2(WMv, WNv, NTv, Vv, KCv)                                                                                 \
    {                                                                                                                  \
        constexpr int BM = 32 * WMv, BN = 32 * NTv * WNv;                                                              \
        const size_t lds = unit_chain2_lds(a.bf, BM);                                                                  \
        if (lds > 160 * 1024) return false;                                                                            \
        if (dry) return true;                                                                                          \
        static unsigned long long attr = 0;                                                                            \
        if (attr_pending(attr)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(unit_chain2_kernel<WMv, WNv, NTv, Vv, KCv>), \
                                               hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); }              \
        set_last_kernel_name("unit_chain2_kernel<" #WMv "," #WNv "," #NTv "," #Vv "," #KCv ">");                        \
        hipLaunchKernelGGL((unit_chain2_kernel<WMv, WNv, NTv, Vv, KCv>), dim3(xcd_grid((a.M + BM - 1) / BM)), dim3(64 * WMv * WNv), lds, s, a); \
        return true;                                                                                                   \
    }
    static const int chain_v = getenv("YN_CHAIN_V") ? atoi(getenv("YN_CHAIN_V")) : 2;      // 1: the round-2 kernel (A/B runs); 2: unit_chain2_kernel
    if (a.Ws2h && chain_v >= 2) {
        // round 5: the persistent, software-pipelined form where it applies (kernels_pipe.hip); coverage is still decided by the tiles below
        if (!dry && launch_unit_pipe(a, s)) return true;
        if (a.Npad == 64 && !v4) YN_UC2(2, 2, 1, 2, 4)
        if (a.Npad == 64 && v4) YN_UC2(2, 2, 1, 4, 4)
        // small maps (one image): 32-row tiles - twice the workgroups, and a workgroup's serial chain (one depthwise round instead of two,
        // half the epilogue rows) is what a launch of a few dozen workgroups costs
        static const int small_m = getenv("YN_CHAIN_SMALL_M") ? atoi(getenv("YN_CHAIN_SMALL_M")) : 4096;
        if (a.Npad == 128 && v4 && a.M <= small_m) YN_UC2(1, 4, 1, 4, 4)
        if (a.Npad == 128 && v4) YN_UC2(2, 2, 2, 4, 3)
        if (a.Npad == 256 && v4) YN_UC2(1, 4, 2, 4, 3)          // 32-row tiles, four wavefronts x 64 columns (NT = 4 would need 128 accumulator + 64 pass-through registers)
        if (a.Npad == 32 && v4) YN_UC2(4, 1, 1, 4, 4)
        if (a.Npad == 96 && v4) YN_UC2(4, 1, 3, 4, 2)
        return false;
    }
#undef YN_UC2
    if (a.Ws2h) {                                            // split-f16 family
        if (a.Npad == 64 && !v4) YN_UCS(2, 2, 1, 2)
        if (a.Npad == 64 && v4) YN_UCS(2, 2, 1, 4)
        if (a.Npad == 128 && v4) YN_UCS(2, 2, 2, 4)
        if (a.Npad == 256 && v4) YN_UCS(2, 2, 4, 4)
        if (a.Npad == 32 && v4) YN_UCS(4, 1, 1, 4)
        if (a.Npad == 96 && v4) YN_UCS(4, 1, 3, 4)
        return false;
    }
#undef YN_UCS
    const int tiles64 = (a.M + 63) / 64;
    if (a.Npad == 64 && !v4) YN_UC(2, 2, 1, 2)
    if (a.Npad == 64 && v4) YN_UC(2, 2, 1, 4)
    if (a.Npad == 128 && v4) YN_UC(2, 2, 2, 4)                 // (a 32-row tile, <1,4,1,4>, measured the same: 42.4 vs 41.4 us)
    if (a.Npad == 256 && v4) { if (tiles64 >= 256) YN_UC(2, 2, 4, 4) else YN_UC(1, 4, 2, 4) }
    if (a.Npad == 32 && v4) YN_UC(4, 1, 1, 4)
    if (a.Npad == 96 && v4) YN_UC(4, 1, 3, 4)
#undef YN_UC
    return false;
}
// -------------------------------------------------------------------------------------------------
// A STRIDE-2 ShuffleV2 unit (backbone/shufflenetv2.py:30-51, 73-74) as one kernel:
//     branch 2:  y1 = relu(pw1(x)) at H x W  ->  y2 = dw3x3_s2(y1) at H/2 x W/2  ->  y3 = relu(pw2(y2))
//     branch 1:  z1 = dw3x3_s2(x)  ->  z2 = relu(pw(z1))                                  out = shuffle(cat(z2, y3))
// y1 is the largest tensor of the network (stage 2: 104 x 104 x 58 per image, 80 MB per 32-image step written and read back once) and the
// launches around it cost their full duration even with four streams (tools/ablate.sh: the big-tensor regions do not overlap with
// anything).  One workgroup = an 8 x 4 tile of OUTPUT pixels of one image: pw1 on the 17 x 9 input pixels the tile's depthwise
// windows cover (20 % recomputed on tile borders; K = cin <= 32 is one chunk), y1 in an fp32 LDS tile (zero outside the image: the
// depthwise conv pads its INPUT), depthwise -> split planes, pw2 (K = bf <= 64: two chunks) on wavefronts 0..NP-1 while wavefronts
// NP..2NP-1 run branch 1's pointwise conv (its depthwise inputs come straight from global into registers at kernel start), and the
// concat + shuffle is the store.  x is read once (x 1.2) and `out` written once: 80 MB per step instead of 360 MB in five launches.
// Every sum runs in the order of gemm_split_kernel / dwconv3x3_kernel: bit-identical to the five launches
// (test_down_unit_is_bit_identical).  LDS 76 KB: two workgroups per CU.  94 us of launches -> 75 us; 33.3 -> 35.2 k images/s together
// with the branch-free activation this kernel led to (three branches per accumulator value in the first version's epilogue).
// pass != null: branch 1's output is read from memory instead (the two-kernel form of branch 1; A/B runs, YN_DOWN_B1=0).
// -------------------------------------------------------------------------------------------------
template <int NP>                                           // Npad / 32 of both GEMMs (bf <= 32 * NP)
__global__ __launch_bounds__(256, 2) void down_unit_kernel(DownArgs a)
{
    constexpr int TW = 8, TH = 4, WW = 2 * TW + 1, WH = 2 * TH + 1, NPIX = WW * WH, RT1 = (NPIX + 31) / 32;      // 17 x 9 = 153 window pixels, 5 row tiles
    constexpr int BN = 32 * NP, AST1 = 32 + 8, AST2 = BN + 8, NO = TW * TH;
    extern __shared__ __attribute__((aligned(16))) float du_smem[];
    const int bf = a.bf, CS = bf + 2;
    // region 1: A1 planes [RT1*32][AST1] x 2 + B1 [4][BN][8] x 2   (GEMM 1), later A2 planes [32][AST2] x 2 + B2 [8][BN][8] x 2 (GEMM 2)
    uch16* A1h = reinterpret_cast<uch16*>(du_smem);
    uch16* A1l = A1h + RT1 * 32 * AST1;
    uch16* B1 = A1l + RT1 * 32 * AST1;                      // hi plane, then lo plane
    constexpr int R1_HALVES_A = 2 * RT1 * 32 * AST1 + 2 * 4 * BN * 8;
    constexpr int R1_HALVES_B = 2 * NO * AST2 + 2 * 8 * BN * 8 + 2 * NO * AST1 + 2 * 4 * BN * 8;      // A2, B2, then branch 1's A3 [32][AST1] x 2 and B3 [4][BN][8] x 2
    constexpr int R1_HALVES = R1_HALVES_A > R1_HALVES_B ? R1_HALVES_A : R1_HALVES_B;
    uch16* A2h = reinterpret_cast<uch16*>(du_smem);
    uch16* A2l = A2h + NO * AST2;
    uch16* B2 = A2l + NO * AST2;
    uch16* A3h = B2 + 2 * 8 * BN * 8;
    uch16* A3l = A3h + NO * AST1;
    uch16* B3 = A3l + NO * AST1;
    const bool fuse1 = a.pass == nullptr;                   // branch 1 (depthwise stride 2 on x, then pointwise) computed here too
    float* T32 = du_smem + (R1_HALVES + 1) / 2;             // [RT1*32][CS] (rows >= NPIX are written as zeros, never read)

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int Ho = a.H >> 1, Wo = a.W >> 1;
    const int tx_n = (Wo + TW - 1) / TW, ty_n = (Ho + TH - 1) / TH;
    const int tile = (int)xcd_block(blockIdx.x, gridDim.x);
    if (tile >= a.B * ty_n * tx_n) return;
    const int b = tile / (ty_n * tx_n), trem = tile - b * (ty_n * tx_n);
    const int oy0 = (trem / tx_n) * TH, ox0 = (trem % tx_n) * TW;
    const int iy0 = 2 * oy0 - 1, ix0 = 2 * ox0 - 1;         // input pixel of window position (0, 0)
    const int KQ1 = (a.cin + 7) >> 3, KQ2 = (bf + 7) >> 3;
    float amax = 0.0f;                                      // range guard (yn_device.h): largest |value| this thread has split

#ifdef YN_EXP_TIMING
    long long TS[8]; int tsn = 0;
#define YN_TS() TS[tsn++] = __builtin_readcyclecounter()
#else
#define YN_TS()
#endif
    YN_TS();
    // ---- 1. x window -> split planes; W1 -> LDS; W2, the depthwise taps and biases -> registers --------------------------------------
    unsigned char* inside = reinterpret_cast<unsigned char*>(T32 + RT1 * 32 * CS);      // [RT1*32] window pixel lies inside the image
    // depthwise work split: thread = (channel pair cp, pixel lane pl); its nine taps and bias stay in registers
    const int cp_n = bf >> 1, ppl = 256 / cp_n;
    const int cp = t % cp_n, dpl = t / cp_n, dc = cp * 2;
    const bool dworker = dpl < ppl;
    // biases of both GEMMs for this lane's columns: requested now, used after the MFMAs (a load inside the epilogue is a full wait)
    float bias1[NP], bias2v = 0.0f;
#pragma unroll
    for (int nt = 0; nt < NP; ++nt) bias1[nt] = (nt * 32 + l31 < bf) ? a.b1[nt * 32 + l31] : 0.0f;
    if (wave < NP && wave * 32 + l31 < bf) bias2v = a.b2[wave * 32 + l31];
    float2 wd[9], bd = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 9; ++k) wd[k] = dworker ? *reinterpret_cast<const float2*>(a.wdw + k * bf + dc) : make_float2(0.0f, 0.0f);
    if (dworker) bd = *reinterpret_cast<const float2*>(a.bdw + dc);
    // branch 1: thread = (channel pair c1, pixel lane p1) -> its <= 2 output pixels' 3x3 stride-2 windows of x, straight from global (the
    // lines are the ones the window load below brings in), nine taps and the bias in registers; W3 -> registers
    const int c1_n = a.cin >> 1, p1_n = 256 / c1_n;
    const int c1 = (t % c1_n) * 2, p1 = t / c1_n;
    constexpr int NI1 = 2;                                  // 32 output pixels over >= 16 pixel lanes (cin <= 32)
    float2 x1w[NI1][9], w1d[9], b1d = make_float2(0.0f, 0.0f);
    constexpr int B3_PER = (2 * 4 * BN + 255) / 256;
    uch16x8 b3_reg[B3_PER];
    float bias3v = 0.0f;
    if (fuse1) {
#pragma unroll
        for (int k = 0; k < 9; ++k) w1d[k] = *reinterpret_cast<const float2*>(a.wdw1 + k * a.cin + c1);
        b1d = *reinterpret_cast<const float2*>(a.bdw1 + c1);
#pragma unroll
        for (int i = 0; i < NI1; ++i) {
            const int op = p1 + i * p1_n;
            const int dy = op / TW, dx = op - dy * TW;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int iy = 2 * (oy0 + dy) - 1 + k / 3, ix = 2 * (ox0 + dx) - 1 + k % 3;
                const bool ok = op < NO && p1 < p1_n && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
                x1w[i][k] = vmask(*reinterpret_cast<const float2*>(a.x + ((size_t)(b * a.H + (ok ? iy : 0)) * a.W + (ok ? ix : 0)) * a.cin + c1), opaque_mask(ok));
            }
        }
#pragma unroll
        for (int i = 0; i < B3_PER; ++i) {
            const int g = t + 256 * i;
            const int pl = g / (4 * BN), r = g - pl * (4 * BN);
            const int o = r / BN, n = r - o * BN;
            uch16x8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (uch16)0.0f;
            if (g < 2 * 4 * BN && o < KQ1 && n < a.Npad3) v = *reinterpret_cast<const uch16x8*>(reinterpret_cast<const uch16*>(pl ? a.W3l : a.W3h) + ((size_t)o * a.Npad3 + n) * 8);
            b3_reg[i] = v;
        }
        if (wave >= NP && wave < 2 * NP && (wave - NP) * 32 + l31 < bf) bias3v = a.b3[(wave - NP) * 32 + l31];
    }
    if (t < RT1 * 32) {                                     // one window pixel per thread: all of its (<= 32) input channels
        const int p = t;
        const int wy = p / WW, wx = p - wy * WW;
        const int iy = iy0 + wy, ix = ix0 + wx;
        const bool ok = p < NPIX && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
        inside[p] = ok ? 1 : 0;
        const float* px = a.x + ((size_t)(b * a.H + (ok ? iy : 0)) * a.W + (ok ? ix : 0)) * a.cin;
        float2 v[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool kj = 2 * j < a.cin;
            v[j] = vmask(*reinterpret_cast<const float2*>(px + (kj ? 2 * j : 0)), opaque_mask(ok && kj));
            amax = range_track(range_track(amax, v[j].x), v[j].y);
        }
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            uch16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x0 = v[o * 4 + j].x, x1 = v[o * 4 + j].y;
                hi[2 * j] = (uch16)x0; hi[2 * j + 1] = (uch16)x1;
                lo[2 * j] = (uch16)((x0 - (float)hi[2 * j]) * 2048.0f); lo[2 * j + 1] = (uch16)((x1 - (float)hi[2 * j + 1]) * 2048.0f);
            }
            *reinterpret_cast<uch16x8*>(A1h + p * AST1 + o * 8) = hi;
            *reinterpret_cast<uch16x8*>(A1l + p * AST1 + o * 8) = lo;
        }
    }
    for (int g = t; g < 2 * 4 * BN; g += 256) {             // W1: plane, octet, column
        const int pl = g / (4 * BN), r = g - pl * (4 * BN);
        const int o = r / BN, n = r - o * BN;
        uch16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (uch16)0.0f;
        if (o < KQ1 && n < a.Npad1) v = *reinterpret_cast<const uch16x8*>(reinterpret_cast<const uch16*>(pl ? a.W1l : a.W1h) + ((size_t)o * a.Npad1 + n) * 8);
        *reinterpret_cast<uch16x8*>(B1 + (size_t)g * 8) = v;
    }
    constexpr int B2_PER = (2 * 8 * BN + 255) / 256;
    uch16x8 b2_reg[B2_PER];
#pragma unroll
    for (int i = 0; i < B2_PER; ++i) {
        const int g = t + 256 * i;
        const int pl = g / (8 * BN), r = g - pl * (8 * BN);
        const int o = r / BN, n = r - o * BN;
        uch16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (uch16)0.0f;
        if (g < 2 * 8 * BN && o < KQ2 && n < a.Npad2) v = *reinterpret_cast<const uch16x8*>(reinterpret_cast<const uch16*>(pl ? a.W2l : a.W2h) + ((size_t)o * a.Npad2 + n) * 8);
        b2_reg[i] = v;
    }
    __syncthreads();
    YN_TS();

    // ---- 2. y1 = act(pw1) on the window pixels -> T32 (zero outside the image): RT1 x NP (row tile, 32-column tile) items over the four
    //      wavefronts (10 items: 3, 3, 2, 2 - whole row tiles would be 2, 1, 1, 1 with twice the work each) -----------------------------
    for (int it = wave; it < RT1 * NP; it += 4) {
        const int rt = it / NP, nt = it - rt * NP;
        f32x16 acc0, acc1;
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc0[k] = 0.0f; acc1[k] = 0.0f; }
        const uch16* Ahb = A1h + (rt * 32 + l31) * AST1 + h * 8;
        const uch16* Alb = A1l + (rt * 32 + l31) * AST1 + h * 8;
        const uch16* Bhb = B1 + (size_t)(h * BN + nt * 32 + l31) * 8;
        const uch16* Blb = Bhb + 4 * BN * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uch16x8 ah = *reinterpret_cast<const uch16x8*>(Ahb + ks * 16);
            const uch16x8 al = *reinterpret_cast<const uch16x8*>(Alb + ks * 16);
            const uch16x8 bh = *reinterpret_cast<const uch16x8*>(Bhb + (size_t)(ks * 2 * BN) * 8);
            const uch16x8 bl = *reinterpret_cast<const uch16x8*>(Blb + (size_t)(ks * 2 * BN) * 8);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1, 0, 0, 0);
        }
        unsigned in16 = 0;                                  // inside flags of this lane's 16 rows
#pragma unroll
        for (int r = 0; r < 16; ++r) in16 |= (unsigned)inside[rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h] << r;
        const int n = nt * 32 + l31;
        float bias = 0.0f;
#pragma unroll
        for (int q = 0; q < NP; ++q) bias = (q == nt) ? bias1[q] : bias;
        if (n < bf) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = rt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const float v = apply_act(__builtin_fmaf(acc1[r], 1.0f / 2048.0f, acc0[r]) + bias, a.act1);
                T32[p * CS + n] = __uint_as_float(__float_as_uint(v) & (0u - ((in16 >> r) & 1u)));
            }
        }
    }
    __syncthreads();                                        // y1 complete; region 1 is free
    YN_TS();

    // ---- 3. depthwise 3x3 stride 2 (dwconv3x3_kernel's fma chain) -> split planes A2; W2 -> LDS ---------------------------------
#pragma unroll
    for (int i = 0; i < B2_PER; ++i) {
        const int g = t + 256 * i;
        if (g < 2 * 8 * BN) *reinterpret_cast<uch16x8*>(B2 + (size_t)g * 8) = b2_reg[i];
    }
    if (fuse1) {
#pragma unroll
        for (int i = 0; i < B3_PER; ++i) {
            const int g = t + 256 * i;
            if (g < 2 * 4 * BN) *reinterpret_cast<uch16x8*>(B3 + (size_t)g * 8) = b3_reg[i];
        }
        if (p1 < p1_n) {
#pragma unroll
            for (int i = 0; i < NI1; ++i) {
                const int op = p1 + i * p1_n;
                if (op < NO) {
                    float2 acc = b1d;
#pragma unroll
                    for (int k = 0; k < 9; ++k) vfma(acc, x1w[i][k], w1d[k]);
                    acc = vact(acc, a.dw1_act);
                    amax = range_track(range_track(amax, acc.x), acc.y);
                    uch16x2 hi, lo;
                    hi[0] = (uch16)acc.x; hi[1] = (uch16)acc.y;
                    lo[0] = (uch16)((acc.x - (float)hi[0]) * 2048.0f); lo[1] = (uch16)((acc.y - (float)hi[1]) * 2048.0f);
                    *reinterpret_cast<uch16x2*>(A3h + op * AST1 + c1) = hi;
                    *reinterpret_cast<uch16x2*>(A3l + op * AST1 + c1) = lo;
                }
            }
        }
        const int pad1 = AST1 - a.cin;                      // K tail of branch 1's planes: zero
        for (int i = t; i < NO * pad1; i += 256) { const int r = i / pad1, c2 = a.cin + i - r * pad1; A3h[r * AST1 + c2] = (uch16)0.0f; A3l[r * AST1 + c2] = (uch16)0.0f; }
    }
    if (dworker) {
        for (int op = dpl; op < NO; op += ppl) {
            const int dy = op / TW, dx = op - dy * TW;
            float2 acc = bd;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx)
                    vfma(acc, *reinterpret_cast<const float2*>(T32 + ((2 * dy + ky) * WW + 2 * dx + kx) * CS + dc), wd[ky * 3 + kx]);
            acc = vact(acc, a.dw_act);
            amax = range_track(range_track(amax, acc.x), acc.y);
            uch16x2 hi, lo;
            hi[0] = (uch16)acc.x; hi[1] = (uch16)acc.y;
            lo[0] = (uch16)((acc.x - (float)hi[0]) * 2048.0f); lo[1] = (uch16)((acc.y - (float)hi[1]) * 2048.0f);
            *reinterpret_cast<uch16x2*>(A2h + op * AST2 + dc) = hi;
            *reinterpret_cast<uch16x2*>(A2l + op * AST2 + dc) = lo;
        }
    }
    {
        const int padn = AST2 - bf;                         // K tail of both planes: zero
        for (int i = t; i < NO * padn; i += 256) { const int r = i / padn, c2 = bf + i - r * padn; A2h[r * AST2 + c2] = (uch16)0.0f; A2l[r * AST2 + c2] = (uch16)0.0f; }
    }
    __syncthreads();
    YN_TS();
    range_report(a.ovf, amax);                              // every split of this workgroup is done

    // ---- 4. branch 1's pointwise conv on wavefronts NP..2NP-1 (-> an LDS tile in the free T32 space) while wavefronts 0..NP-1 run pw2 --
    float* PT = T32;                                        // [32][BN + 1]
    f32x16 acc0, acc1;
#pragma unroll
    for (int k = 0; k < 16; ++k) { acc0[k] = 0.0f; acc1[k] = 0.0f; }
    if (fuse1 && wave >= NP && wave < 2 * NP) {
        const int nt = wave - NP;
        const uch16* Ahb = A3h + l31 * AST1 + h * 8;
        const uch16* Alb = A3l + l31 * AST1 + h * 8;
        const uch16* Bhb = B3 + (size_t)(h * BN + nt * 32 + l31) * 8;
        const uch16* Blb = Bhb + 4 * BN * 8;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const uch16x8 ah = *reinterpret_cast<const uch16x8*>(Ahb + ks * 16);
            const uch16x8 al = *reinterpret_cast<const uch16x8*>(Alb + ks * 16);
            const uch16x8 bh = *reinterpret_cast<const uch16x8*>(Bhb + (size_t)(ks * 2 * BN) * 8);
            const uch16x8 bl = *reinterpret_cast<const uch16x8*>(Blb + (size_t)(ks * 2 * BN) * 8);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1, 0, 0, 0);
        }
        const int n = nt * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r)
            PT[((r & 3) + 8 * (r >> 2) + 4 * h) * (BN + 1) + n] = apply_act(__builtin_fmaf(acc1[r], 1.0f / 2048.0f, acc0[r]) + bias3v, a.act3);
    }
    if (wave < NP) {                                        // pw2: one wavefront per 32 output columns
        const int nt = wave;
        const uch16* Ahb = A2h + l31 * AST2 + h * 8;
        const uch16* Alb = A2l + l31 * AST2 + h * 8;
        const uch16* Bhb = B2 + (size_t)(h * BN + nt * 32 + l31) * 8;
        const uch16* Blb = Bhb + 8 * BN * 8;
        for (int ks = 0; ks < 2 * ((KQ2 + 3) >> 2); ++ks) {                 // gemm_split_tile's chunks of 32: whole chunks, zero-padded
            const uch16x8 ah = *reinterpret_cast<const uch16x8*>(Ahb + ks * 16);
            const uch16x8 al = *reinterpret_cast<const uch16x8*>(Alb + ks * 16);
            const uch16x8 bh = *reinterpret_cast<const uch16x8*>(Bhb + (size_t)(ks * 2 * BN) * 8);
            const uch16x8 bl = *reinterpret_cast<const uch16x8*>(Blb + (size_t)(ks * 2 * BN) * 8);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1, 0, 0, 0);
        }
    }
    if (fuse1) __syncthreads();                             // branch 1's tile is complete
    // ---- 5. concat + shuffle store: out[2n] = branch 1, out[2n+1] = branch 2 -------------------------------------------------------
    if (wave < NP) {
        const int n = wave * 32 + l31;
        if (n < bf) {
            const float bias = bias2v;
            // the 16 pass-through values of this lane are requested together, before any store (a load issued next to the store that
            // needs it is followed by a full wait: 16 memory latencies in a row)
            float pv[16];
            size_t mrow[16];
            unsigned okm = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int op = (r & 3) + 8 * (r >> 2) + 4 * h;
                const int oy = oy0 + op / TW, ox = ox0 + op % TW;
                const bool ok = oy < Ho && ox < Wo;
                okm |= (ok ? 1u : 0u) << r;
                mrow[r] = ((size_t)b * Ho + (ok ? oy : 0)) * Wo + (ok ? ox : 0);
                pv[r] = fuse1 ? PT[op * (BN + 1) + n] : __uint_as_float(__float_as_uint(a.pass[mrow[r] * bf + n]) & opaque_mask(ok));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                if ((okm >> r) & 1u) {
                    const float v = apply_act(__builtin_fmaf(acc1[r], 1.0f / 2048.0f, acc0[r]) + bias, a.act2);
                    *reinterpret_cast<float2*>(a.out + mrow[r] * (2 * bf) + 2 * n) = make_float2(pv[r], v);
                }
            }
        }
    }
#ifdef YN_EXP_TIMING
    YN_TS();
    if (t == 0 && (blockIdx.x % 331) == 7) printf("downunit blk %d load+split %lld gemm1 %lld dw %lld gemm2+store %lld total %lld\n", (int)blockIdx.x, TS[1] - TS[0], TS[2] - TS[1], TS[3] - TS[2], TS[4] - TS[3], TS[4] - TS[0]);
#endif
#undef YN_TS
}

// -------------------------------------------------------------------------------------------------
// down_unit_kernel as a software pipeline (round 4).  `tools/phase_timing.sh downunit` on the one-tile-per-workgroup form above: 25 k
// cycles per workgroup, 10 k of them the load phase (one memory round trip, 34 eight-byte loads per thread with a cache line per lane),
// at two workgroups of four wavefronts per CU - nothing to hide it behind.  This form keeps the arithmetic and its order (same bits,
// test_down_unit_is_bit_identical) and changes where the operands come from:
//   * a workgroup WALKS tiles (XCD-contiguous: workgroup i of XCD x takes tiles x*TL + i, + G/8, ...) and requests the next tile's input
//     window right after the barrier that frees the registers of the current one: the round trip runs under phases 2-5;
//   * the window is loaded as what it is - nine contiguous runs of 17 pixels x cin floats - with 16-byte loads (4-5 per thread, eight
//     cache lines per wavefront instruction instead of 64), split into the A planes AND kept as fp32 in LDS: branch 1's depthwise conv
//     reads its stride-2 windows there (18 more global loads per thread in the old form);
//   * every weight is loop-invariant: the B fragments of a wavefront's columns (16 contiguous bytes of the pre-split pack per lane and
//     k-step - down2_kernel's register-direct form) are loaded ONCE per workgroup into 48 registers, like the depthwise taps.  No
//     weight ever passes through LDS: 76 -> 78 KB with the fp32 window (two workgroups per CU either way).
// -------------------------------------------------------------------------------------------------
template <int NP, bool RELU>                                 // RELU: the three pointwise convs end in ReLU, the depthwise convs in nothing (ShuffleNetV2)
__global__ __launch_bounds__(256, 2) void down_unit_pipe_kernel(DownArgs a, int tiles)
{
    constexpr int TW = 8, TH = 4, WW = 2 * TW + 1, WH = 2 * TH + 1, NPIX = WW * WH, RT1 = (NPIX + 31) / 32;
    constexpr int BN = 32 * NP, AST1 = 32 + 8, AST2 = BN + 8, NO = TW * TH, NLD = 5, KS2 = 2 * NP;
    extern __shared__ __attribute__((aligned(16))) float du_smem[];
    const int bf = a.bf, CS = bf + 2, cin = a.cin, cq = cin >> 2;
    uch16* A1h = reinterpret_cast<uch16*>(du_smem);                  // [RT1*32][AST1] x 2: pw1's operand planes
    uch16* A1l = A1h + RT1 * 32 * AST1;
    uch16* A2h = A1h;                                                // behind GEMM 1: [NO][AST2] x 2 (pw2), then [NO][AST1] x 2 (branch 1)
    uch16* A2l = A2h + NO * AST2;
    uch16* A3h = A2l + NO * AST2;
    uch16* A3l = A3h + NO * AST1;
    float* X32 = du_smem + RT1 * 32 * AST1;                          // [RT1*32][cin]: the window in fp32 (zero outside the image)
    float* T32 = X32 + RT1 * 32 * cin;                               // [RT1*32][CS]: y1 (zero outside the image), later branch 1's tile
    unsigned char* inside = reinterpret_cast<unsigned char*>(T32 + RT1 * 32 * CS);      // [RT1*32]

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, h = lane >> 5;
    const int Ho = a.H >> 1, Wo = a.W >> 1;
    const int tx_n = (Wo + TW - 1) / TW, ty_n = (Ho + TH - 1) / TH, per_img = tx_n * ty_n;
    const int KQ1 = (cin + 7) >> 3, KQ2 = (bf + 7) >> 3;
    // the walk: XCD x = blockIdx & 7 owns tiles [x * TL, (x + 1) * TL)
    const int TL = (tiles + 7) >> 3, GL = (int)(gridDim.x >> 3);
    const int tbase = (int)(blockIdx.x & 7u) * TL;
    int tl = (int)(blockIdx.x >> 3);
    if (tl >= TL || tbase + tl >= tiles) return;
    auto act_pw = [&](float v, int act) { return RELU ? __builtin_fmaxf(v, 0.0f) : apply_act(v, act); };

#ifdef YN_EXP_TIMING
    long long TS[8]; int tsn = 0;
#define YN_TS() if (tsn < 8) TS[tsn++] = __builtin_readcyclecounter()
#else
#define YN_TS()
#endif
    YN_TS();
    // ---- loop invariants: depthwise taps and biases of both branches, the B fragments of this wavefront's columns --------------------
    const int cp_n = bf >> 1, ppl = 256 / cp_n;
    const int cp = t % cp_n, dpl = t / cp_n, dc = cp * 2;
    const bool dworker = dpl < ppl;
    float2 wd[9], bd = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int k = 0; k < 9; ++k) wd[k] = dworker ? *reinterpret_cast<const float2*>(a.wdw + k * bf + dc) : make_float2(0.0f, 0.0f);
    if (dworker) bd = *reinterpret_cast<const float2*>(a.bdw + dc);
    const int c1_n = cin >> 1, p1_n = 256 / c1_n;
    const int c1 = (t % c1_n) * 2, p1 = t / c1_n;
    constexpr int NI1 = 2;                                           // 32 output pixels over >= 16 pixel lanes (cin <= 32)
    float2 w1d[9], b1d;
#pragma unroll
    for (int k = 0; k < 9; ++k) w1d[k] = *reinterpret_cast<const float2*>(a.wdw1 + k * cin + c1);
    b1d = *reinterpret_cast<const float2*>(a.bdw1 + c1);
    // GEMM 1: items it = (row tile, column tile) = 3 - wave, 7 - wave, ... - the wavefronts that run pw2 and the stores take the smaller share
    const int it0 = 3 - wave;
    const int nt1 = (NP == 2) ? (it0 & 1) : 0;                       // the stride 4 is even: a wavefront's column tile never changes
    const bool pw2_wave = wave < NP, b1_wave = wave >= NP && wave < 2 * NP;
    const int nt2 = pw2_wave ? wave : wave - NP;
    auto frag = [&](const void* Wh_, const void* Wl_, int ks, int KQ, int n, bool use, uch16x8& bh, uch16x8& bl) {
        const int kq = ks * 2 + h;
        const bool ok = use && kq < KQ;
        const size_t off = ((size_t)(ok ? kq : 0) * BN + (ok ? n : 0)) * 8;      // Npad1 = Npad2 = Npad3 = BN (down_unit_covers)
        const unsigned mk = opaque_mask(ok);
        uint4 vh = *reinterpret_cast<const uint4*>(reinterpret_cast<const uch16*>(Wh_) + off), vl = *reinterpret_cast<const uint4*>(reinterpret_cast<const uch16*>(Wl_) + off);
        vh.x &= mk; vh.y &= mk; vh.z &= mk; vh.w &= mk; vl.x &= mk; vl.y &= mk; vl.z &= mk; vl.w &= mk;
        bh = *reinterpret_cast<uch16x8*>(&vh); bl = *reinterpret_cast<uch16x8*>(&vl);
    };
    uch16x8 g1h[2], g1l[2], g2h[KS2], g2l[KS2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) frag(a.W1h, a.W1l, ks, KQ1, nt1 * 32 + l31, true, g1h[ks], g1l[ks]);
#pragma unroll
    for (int ks = 0; ks < KS2; ++ks) {                               // pw2 (K = bf) on wavefronts 0..NP-1, branch 1's pointwise conv (K = cin: two steps) behind them
        const bool use = pw2_wave || (b1_wave && ks < 2);
        frag(pw2_wave ? a.W2h : a.W3h, pw2_wave ? a.W2l : a.W3l, ks, pw2_wave ? KQ2 : KQ1, nt2 * 32 + l31, use, g2h[ks], g2l[ks]);
    }
    const float bias1 = (nt1 * 32 + l31 < bf) ? a.b1[nt1 * 32 + l31] : 0.0f;
    const float bias23 = (nt2 * 32 + l31 < bf && (pw2_wave || b1_wave)) ? (pw2_wave ? a.b2 : a.b3)[nt2 * 32 + l31] : 0.0f;
    // window pieces of this thread: 16-byte piece e = t + 256 i of the WH runs of WW * cin floats; everything about a piece that does not
    // depend on the tile is computed here
    const int QR = WW * cq;
    int wpos[NLD], wrel[NLD];                                        // wy | wx << 8 | first channel << 16 (or -1); offset from the window's first float
#pragma unroll
    for (int i = 0; i < NLD; ++i) {
        const int e = t + 256 * i;
        const int wy = e / QR, q = e - wy * QR;
        const int wx = q / cq, c4 = (q - wx * cq) * 4;
        wpos[i] = (e < WH * QR) ? (wy | (wx << 8) | (c4 << 16)) : -1;
        wrel[i] = (wy * a.W + wx) * cin + c4;
    }
    float4 pre[NLD];
    unsigned pok = 0;
    const char* xbase = reinterpret_cast<const char*>(a.x);
    auto request = [&](int tile) {
        const int b = tile / per_img, trem = tile - b * per_img;
        const int ty = trem / tx_n, tx = trem - ty * tx_n;
        const int iy0 = 2 * ty * TH - 1, ix0 = 2 * tx * TW - 1;
        const int org = ((b * a.H + iy0) * a.W + ix0) * cin;         // the window's first float (outside the tensor on the image border: never used then)
        pok = 0;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const int wy = wpos[i] & 0xff, wx = (wpos[i] >> 8) & 0xff;
            const bool ok = wpos[i] >= 0 && (unsigned)(iy0 + wy) < (unsigned)a.H && (unsigned)(ix0 + wx) < (unsigned)a.W;
            // raw value from a clamped address; zeroed when it is CONSUMED (a mask applied here would wait for the load here)
            pre[i] = *reinterpret_cast<const float4*>(xbase + (unsigned)(ok ? org + wrel[i] : 0) * 4u);
            pok |= (ok ? 1u : 0u) << i;
        }
    };
    request(tbase + tl);
    if (t < RT1 * 32 - NPIX) inside[NPIX + t] = 0;                   // pad rows of the last row tile: never inside
    float amax = 0.0f;                                               // range guard (yn_device.h): largest |value| this thread has split
    float* PT = T32;                                                 // [32][BN + 1]: branch 1's output tile

    // ---- 1. the window: fp32 copy + split planes; K tail of the planes (run for the NEXT tile in front of the current tile's stores: the wait
    //      for the prefetch must not stand behind them in the memory counter) --------------------------------------------------------
    auto consume = [&]() {
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            if (wpos[i] >= 0) {
                const int wy = wpos[i] & 0xff, wx = (wpos[i] >> 8) & 0xff, c4 = (wpos[i] >> 16) & 0xff;
                const int p = wy * WW + wx;
                const float4 v = vmask(pre[i], 0u - ((pok >> i) & 1u));
                *reinterpret_cast<float4*>(X32 + p * cin + c4) = v;
                const float x4[4] = {v.x, v.y, v.z, v.w};
                uch16x4 hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) { amax = range_track(amax, x4[j]); hi[j] = (uch16)x4[j]; lo[j] = (uch16)((x4[j] - (float)hi[j]) * 2048.0f); }
                *reinterpret_cast<uch16x4*>(A1h + p * AST1 + c4) = hi;
                *reinterpret_cast<uch16x4*>(A1l + p * AST1 + c4) = lo;
                if (c4 == 0) inside[p] = (unsigned char)((pok >> i) & 1u);
            }
        }
        {
            const int ntail = (32 - cin) >> 2;                       // quads of zero columns up to K = 32
            for (int i = t; i < RT1 * 32 * ntail; i += 256) {
                const int p = i / ntail, c2 = cin + 4 * (i - p * ntail);
                *reinterpret_cast<uint2*>(A1h + p * AST1 + c2) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(A1l + p * AST1 + c2) = make_uint2(0u, 0u);
            }
        }
    };
    consume();

    for (;;) {
        const int tile = tbase + tl;
        const int b = tile / per_img, trem = tile - b * per_img;
        const int oy0 = (trem / tx_n) * TH, ox0 = (trem % tx_n) * TW;
        __syncthreads();
        YN_TS();
        const bool more = tl + GL < TL && tbase + tl + GL < tiles;
        if (more) request(tbase + tl + GL);                          // the next tile's round trip runs under phases 2-4

        // ---- 2. y1 = act(pw1) on the window pixels -> T32 (zero outside the image) ---------------------------------------------------
        for (int it = it0; it < RT1 * NP; it += 4) {
            const int rt = it / NP;
            f32x16 acc0, acc1;
#pragma unroll
            for (int k = 0; k < 16; ++k) { acc0[k] = 0.0f; acc1[k] = 0.0f; }
            const uch16* Ahb = A1h + (rt * 32 + l31) * AST1 + h * 8;
            const uch16* Alb = A1l + (rt * 32 + l31) * AST1 + h * 8;
            int inw[4];                                              // inside flags of this lane's 16 rows: four bytes per word
#pragma unroll
            for (int g = 0; g < 4; ++g) inw[g] = *reinterpret_cast<const int*>(inside + rt * 32 + 8 * g + 4 * h);
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uch16x8 ah = *reinterpret_cast<const uch16x8*>(Ahb + ks * 16);
                const uch16x8 al = *reinterpret_cast<const uch16x8*>(Alb + ks * 16);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, g1h[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, g1l[ks], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, g1h[ks], acc1, 0, 0, 0);
            }
            const int n = nt1 * 32 + l31;
            if (n < bf) {
                float* trow = T32 + (rt * 32 + 4 * h) * CS + n;      // row (r & 3) + 8 (r >> 2) of this lane's half: a wave-uniform offset
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = act_pw(__builtin_fmaf(acc1[r], 1.0f / 2048.0f, acc0[r]) + bias1, a.act1);
                    trow[((r & 3) + 8 * (r >> 2)) * CS] = __uint_as_float(__float_as_uint(v) & (unsigned)__builtin_amdgcn_sbfe(inw[r >> 2], 8 * (r & 3), 1));
                }
            }
        }
        __syncthreads();                                             // y1 complete; the A1 planes are free
        YN_TS();

        // ---- 3. both depthwise convs (stride 2, dwconv3x3_kernel's fma chain) -> the split planes of their branch --------------------
        if (p1 < p1_n) {
#pragma unroll
            for (int i = 0; i < NI1; ++i) {
                const int op = p1 + i * p1_n;
                if (op < NO) {
                    const int dy = op / TW, dx = op - dy * TW;
                    float2 acc = b1d;
                    const float* xw = X32 + (2 * dy * WW + 2 * dx) * cin + c1;
#pragma unroll
                    for (int k = 0; k < 9; ++k) vfma(acc, *reinterpret_cast<const float2*>(xw + ((k / 3) * WW + k % 3) * cin), w1d[k]);
                    if (!RELU) acc = vact(acc, a.dw1_act);
                    amax = range_track(range_track(amax, acc.x), acc.y);
                    uch16x2 hi, lo;
                    hi[0] = (uch16)acc.x; hi[1] = (uch16)acc.y;
                    lo[0] = (uch16)((acc.x - (float)hi[0]) * 2048.0f); lo[1] = (uch16)((acc.y - (float)hi[1]) * 2048.0f);
                    *reinterpret_cast<uch16x2*>(A3h + op * AST1 + c1) = hi;
                    *reinterpret_cast<uch16x2*>(A3l + op * AST1 + c1) = lo;
                }
            }
        }
        {
            const int pad1 = (AST1 - cin) >> 2;                      // K tail of branch 1's planes: zero (quads: cin is a multiple of 4)
            for (int i = t; i < NO * pad1; i += 256) {
                const int r = i / pad1, c2 = cin + 4 * (i - r * pad1);
                *reinterpret_cast<uint2*>(A3h + r * AST1 + c2) = make_uint2(0u, 0u);
                *reinterpret_cast<uint2*>(A3l + r * AST1 + c2) = make_uint2(0u, 0u);
            }
        }
        if (dworker) {
            for (int op = dpl; op < NO; op += ppl) {
                const int dy = op / TW, dx = op - dy * TW;
                float2 acc = bd;
                const float* yw = T32 + (2 * dy * WW + 2 * dx) * CS + dc;
#pragma unroll
                for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx)
                        vfma(acc, *reinterpret_cast<const float2*>(yw + (ky * WW + kx) * CS), wd[ky * 3 + kx]);
                if (!RELU) acc = vact(acc, a.dw_act);
                amax = range_track(range_track(amax, acc.x), acc.y);
                uch16x2 hi, lo;
                hi[0] = (uch16)acc.x; hi[1] = (uch16)acc.y;
                lo[0] = (uch16)((acc.x - (float)hi[0]) * 2048.0f); lo[1] = (uch16)((acc.y - (float)hi[1]) * 2048.0f);
                *reinterpret_cast<uch16x2*>(A2h + op * AST2 + dc) = hi;
                *reinterpret_cast<uch16x2*>(A2l + op * AST2 + dc) = lo;
            }
        }
        {
            const int padn = (AST2 - bf) >> 1;                       // K tail of both planes: zero (pairs: bf is even)
            for (int i = t; i < NO * padn; i += 256) {
                const int r = i / padn, c2 = bf + 2 * (i - r * padn);
                *reinterpret_cast<unsigned*>(A2h + r * AST2 + c2) = 0u;
                *reinterpret_cast<unsigned*>(A2l + r * AST2 + c2) = 0u;
            }
        }
        __syncthreads();
        YN_TS();

        // ---- 4. branch 1's pointwise conv on wavefronts NP..2NP-1 (-> PT, in the free T32 space) while wavefronts 0..NP-1 run pw2 -----
        f32x16 acc0, acc1;
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc0[k] = 0.0f; acc1[k] = 0.0f; }
        if (b1_wave) {
            const uch16* Ahb = A3h + l31 * AST1 + h * 8;
            const uch16* Alb = A3l + l31 * AST1 + h * 8;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const uch16x8 ah = *reinterpret_cast<const uch16x8*>(Ahb + ks * 16);
                const uch16x8 al = *reinterpret_cast<const uch16x8*>(Alb + ks * 16);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, g2h[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, g2l[ks], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, g2h[ks], acc1, 0, 0, 0);
            }
            float* prow = PT + (4 * h) * (BN + 1) + nt2 * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                prow[((r & 3) + 8 * (r >> 2)) * (BN + 1)] = act_pw(__builtin_fmaf(acc1[r], 1.0f / 2048.0f, acc0[r]) + bias23, a.act3);
        }
        if (pw2_wave) {                                              // pw2: one wavefront per 32 output columns, gemm_split_tile's chunks of 32 (whole chunks, zero-padded)
            const uch16* Ahb = A2h + l31 * AST2 + h * 8;
            const uch16* Alb = A2l + l31 * AST2 + h * 8;
#pragma unroll
            for (int ks = 0; ks < KS2; ++ks) {
                const uch16x8 ah = *reinterpret_cast<const uch16x8*>(Ahb + ks * 16);
                const uch16x8 al = *reinterpret_cast<const uch16x8*>(Alb + ks * 16);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, g2h[ks], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, g2l[ks], acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, g2h[ks], acc1, 0, 0, 0);
            }
        }
        __syncthreads();                                             // branch 1's tile is complete; nobody reads the planes, the fp32 window or the flags any more
        if (more) consume();                                         // the next tile's window -> LDS (PT, which the store below reads, lies in T32)
        // ---- 5. concat + shuffle store: out[2n] = branch 1, out[2n+1] = branch 2.  Row r of the accumulator = tile pixel (r >> 2, (r & 3) + 4 h):
        //      a wave-uniform base + a 32-bit lane offset per store -----------------------------------------------------------------------
        if (pw2_wave) {
            const int n = wave * 32 + l31;
            // byte offsets in 32 bits (down_unit_covers: the output tensor is below 4 GB): ONE 64-bit base, the kernel argument itself
            char* obase = reinterpret_cast<char*>(a.out);
            const unsigned lane_off = (unsigned)((((b * Ho + oy0) * Wo + ox0) + 4 * h) * 2 * bf + 2 * n) * 4u;
            const float* prow = PT + (4 * h) * (BN + 1) + n;
            float pv[16];                                            // branch 1's values first: one LDS wait, not sixteen
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = prow[((r & 3) + 8 * (r >> 2)) * (BN + 1)];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dy = r >> 2, dxu = r & 3;
                if (oy0 + dy < Ho) {
                    if (n < bf && ox0 + dxu + 4 * h < Wo) {
                        const float v = act_pw(__builtin_fmaf(acc1[r], 1.0f / 2048.0f, acc0[r]) + bias23, a.act2);
                        *reinterpret_cast<float2*>(obase + (lane_off + (unsigned)((dy * Wo + dxu) * 2 * bf) * 4u)) = make_float2(pv[r], v);
                    }
                }
            }
        }
        YN_TS();
        if (!more) break;
        tl += GL;
    }
    range_report(a.ovf, amax);
#ifdef YN_EXP_TIMING
    if (t == 0 && (blockIdx.x % 61) == 7 && tsn >= 5)
        printf("downpipe blk %d invariants+window %lld gemm1 %lld dw %lld gemm2+store %lld first tile %lld second %lld\n", (int)blockIdx.x, TS[1] - TS[0], TS[2] - TS[1], TS[3] - TS[2],
               TS[4] - TS[3], TS[4] - TS[0], tsn >= 8 ? TS[7] - TS[4] : -1LL);
#endif
#undef YN_TS
}

static size_t down_unit_pipe_lds(int bf, int cin)
{
    return ((size_t)160 * 40 + (size_t)160 * cin + (size_t)160 * (bf + 2)) * sizeof(float) + 160;
}

static size_t down_unit_lds(int bf, int NP)
{
    const int BN = 32 * NP, RT1 = 5, NO = 32;
    const size_t r1a = (size_t)2 * RT1 * 32 * 40 + (size_t)2 * 4 * BN * 8, r1b = (size_t)2 * NO * (BN + 8) + (size_t)2 * 8 * BN * 8 + (size_t)2 * NO * 40 + (size_t)2 * 4 * BN * 8;
    const size_t r1 = r1a > r1b ? r1a : r1b;
    return ((r1 + 1) / 2) * sizeof(float) + (size_t)160 * (bf + 2) * sizeof(float) + 160;       // + the window's inside flags
}

bool down_unit_covers(const DownArgs& a)
{
    return a.W1h && a.W1l && a.W2h && a.W2l && a.cin <= 32 && !(a.cin & 1) && a.bf <= 64 && !(a.bf & 1) && a.Npad1 == a.Npad2 && a.Npad1 <= 64 &&
           !(a.H & 1) && !(a.W & 1) && a.B > 0 && a.cin >= 16 && (size_t)a.B * a.H * a.W * (size_t)(a.cin > a.bf / 2 ? a.cin : a.bf / 2) < ((size_t)1 << 30) &&      // 32-bit byte offsets
           (a.pass || (a.wdw1 && a.bdw1 && a.W3h && a.W3l && a.b3 && a.Npad3 == a.Npad1));
}

void launch_down_unit(const DownArgs& a, hipStream_t s)
{
    const int Ho = a.H >> 1, Wo = a.W >> 1;
    const unsigned tiles = (unsigned)a.B * ((Ho + 3) / 4) * ((Wo + 7) / 8);
    const int NP = a.Npad1 / 32;
    const size_t lds = down_unit_lds(a.bf, NP);
    static unsigned long long attr = 0;
    if (attr_pending(attr)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(down_unit_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(down_unit_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(down_unit_pipe_kernel<1, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(down_unit_pipe_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(down_unit_pipe_kernel<1, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(down_unit_pipe_kernel<2, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    // the pipelined form (a workgroup walks tiles): branch 1 computed here, channel quads (YN_DOWN_PIPE=0: one tile per workgroup, A/B runs)
    static const int pipe = getenv("YN_DOWN_PIPE") ? atoi(getenv("YN_DOWN_PIPE")) : 1;
    static const int pipe_g = getenv("YN_DOWN_PIPE_G") ? atoi(getenv("YN_DOWN_PIPE_G")) : 1024;     // walking workgroups: ~3 tiles each at bs = 32 (512 / 768 / 1024 / 1456 / one per tile measured within 1 % of each other)
    if (pipe && !a.pass && !(a.cin & 3)) {
        const size_t plds = down_unit_pipe_lds(a.bf, a.cin);
        unsigned g = xcd_grid(tiles);
        const unsigned cap = (unsigned)((pipe_g > 8 ? pipe_g : 8) & ~7);
        if (g > cap) g = cap;
        const bool relu = a.act1 == 1 && a.act2 == 1 && a.act3 == 1 && a.dw_act == 0 && a.dw1_act == 0;
#define YN_DUP(np, rl) { set_last_kernel_name("down_unit_pipe_kernel<" #np "," #rl ">"); hipLaunchKernelGGL((down_unit_pipe_kernel<np, rl>), dim3(g), dim3(256), plds, s, a, (int)tiles); }
        if (NP == 1) { if (relu) YN_DUP(1, true) else YN_DUP(1, false) }
        else         { if (relu) YN_DUP(2, true) else YN_DUP(2, false) }
#undef YN_DUP
        return;
    }
    if (NP == 1) { set_last_kernel_name("down_unit_kernel<1>"); hipLaunchKernelGGL(down_unit_kernel<1>, dim3(xcd_grid(tiles)), dim3(256), lds, s, a); }
    else         { set_last_kernel_name("down_unit_kernel<2>"); hipLaunchKernelGGL(down_unit_kernel<2>, dim3(xcd_grid(tiles)), dim3(256), lds, s, a); }
}

// -------------------------------------------------------------------------------------------------
// The stride-2 ShuffleV2 units of stages 3 and 4 (backbone/shufflenetv2.py:42-49, 53-63, 73-74; cin = bf = 116 / 232): too wide for
// down_unit_kernel's window form (pw1 on the 17 x 9 halo needs the whole K = cin of 153 pixels in LDS).  The unit is cut where the chain
// kernels cut theirs - at the depthwise convs, after which everything is pixel-local:
//     launch 1 (gemm_split_kernel):  y1 = relu(pw1(x))                                          [B][H][W][bf]
//     launch 2 (down2_kernel):       y3 = relu(pw2(dw_s2(y1))),  z2 = relu(pw(dw_s2(x))),  out = shuffle(cat(z2, y3))
// Two launches instead of five; the four tensors between them (both depthwise outputs, branch 1's output, 60 MB per step at stage 3)
// never reach memory.  Workgroup = 32 consecutive output pixels (flat, all images): thread = (4 channels, every ppl-th pixel) with the
// nine taps of its channels in registers; the 3 x 3 stride-2 windows of TWO pixels are in flight per round (18 clamped, masked 16-byte
// loads), results split straight into the A planes of their branch; then ONE chunk loop walks the K chunks of pw2 and of branch 1's
// pointwise conv back to back (the next chunk's weights always in flight), four wavefronts x NT column tiles.  The accumulator layout
// gives a lane the SAME column n and the same 16 rows in both GEMMs, so the concat + shuffle is the store: out[row][2n .. 2n+1] =
// (z2, y3) as one 8-byte store.  Every sum in the order of dwconv3x3_kernel / gemm_split_kernel: bit-identical to the five launches
// (test_down_unit_is_bit_identical).  LDS 48 KB at bf = 116 (three workgroups per CU: the 676 workgroups of a 32-image step are all resident).
// -------------------------------------------------------------------------------------------------
template <int NT, int D>                                   // NT column tiles of 32 per wavefront (4 wavefronts); D k-steps of weights in flight
__global__ __launch_bounds__(256, NT == 1 ? 3 : 1) void down2_kernel(Down2Args a)
{
    constexpr int BM = 32;
    extern __shared__ __attribute__((aligned(16))) float d2_smem[];
    const int bf = a.bf, cin = a.cin;
    const int KQ2 = (bf + 7) >> 3, KQ1 = (cin + 7) >> 3, PS2 = plane_stride(bf), PS1 = plane_stride(cin);
    const int S2 = (KQ2 + 1) >> 1, S1 = (KQ1 + 1) >> 1, S = S2 + S1;    // 16-deep k-steps of pw2, of branch 1's pointwise conv
    const bool chain = a.W1nh != nullptr;                               // + the next unit's pw1 (K = bf: S2 steps more, after the store)
    uch16* A2h = reinterpret_cast<uch16*>(d2_smem);                     // [BM][PS2]
    uch16* A2l = A2h + BM * PS2;
    uch16* A1h = A2l + BM * PS2;                                        // [BM][PS1]
    uch16* A1l = A1h + BM * PS1;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int Ho = (a.H - 1) / 2 + 1, Wo = (a.W - 1) / 2 + 1, HWo = Ho * Wo, Mo = a.B * HWo;
    const int m0 = (int)xcd_block(blockIdx.x, gridDim.x) * BM;
    if (m0 >= Mo) return;
    const int nrows = Mo - m0 < BM ? Mo - m0 : BM;

#ifdef YN_EXP_TIMING
    long long TS[8]; int tsn = 0;
#define YN_TS() TS[tsn++] = __builtin_readcyclecounter()
#else
#define YN_TS()
#endif
    YN_TS();
    // The weights never pass through LDS: a lane's B fragment of a k-step is 16 contiguous bytes of the pre-split pack
    // ([octet][column][8 halves]; the 32 lanes of a half-wavefront read 512 contiguous bytes), so every wavefront streams the fragments of
    // ITS columns straight from L2 into registers, D k-steps ahead, with no barrier between the steps of the two pointwise convs (the first
    // form staged K chunks through LDS: eight barrier rounds of ~3.7 k cycles, each waiting for a load issued one round earlier behind the
    // other workgroups' window traffic - 30 k of the workgroup's 60 k cycles).  Step s of the walk: pw2's steps, then branch 1's.
    // The walk is laid out in PADDED steps: every GEMM starts at a multiple of D (G2 = S2 rounded up, G1 likewise), so that the unrolled
    // D-step body holds nothing but loads, two LDS reads and MFMAs, and what happens between the GEMMs stands once, between the loops (it
    // stood inside every unrolled step, D copies of the store epilogue: a 50-100 KB loop body the instruction cache could not hold - a k-step
    // cost ~900 cycles for 192 cycles of MFMAs even with six workgroups on an idle chip).  A padded step multiplies the zero K tail: exact zeros.
    const int G2 = (S2 + D - 1) / D * D, G1 = (S1 + D - 1) / D * D;
    uch16x8 bq[D][NT][2];
    auto load_b = [&](int s, uch16x8 (&dst)[NT][2]) {          // s: padded step
        const bool second = s >= G2, third = s >= G2 + G1;
        const int ks = third ? s - (G2 + G1) : (second ? s - G2 : s), KQ = (second && !third) ? KQ1 : KQ2;
        // (without a next unit the steps >= S are masked loads at a clamped address: the base must still be a real pointer)
        const uch16* Wh = reinterpret_cast<const uch16*>((third && chain) ? a.W1nh : ((second && !third) ? a.W3h : a.W2h));
        const uch16* Wl = reinterpret_cast<const uch16*>((third && chain) ? a.W1nl : ((second && !third) ? a.W3l : a.W2l));
        // no masks (unit_chain2_kernel's load_b: a masked load is waited for where it is issued - the walk then paid a memory round trip per
        // k-step): clamped octet / column, and the steps past the walk take their A fragment from the planes' zero K tail
        const int kq = min(ks * 2 + h, KQ - 1);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = min((wave * NT + nt) * 32 + l31, a.Npad - 1);
            const size_t off = ((size_t)kq * a.Npad + n) * 8;
            dst[nt][0] = *reinterpret_cast<const uch16x8*>(Wh + off);
            dst[nt][1] = *reinterpret_cast<const uch16x8*>(Wl + off);
        }
    };
#pragma unroll
    for (int j = 0; j < D; ++j) load_b(j, bq[j]);                        // requested first: they return before the window loads below
    // ---- 1. both depthwise convs (stride 2, dwconv3x3_kernel's fma chain) -> the split planes of their branch ---------------------------
    float amax = 0.0f;                                                   // range guard (yn_device.h)
    auto dw_branch = [&](const float* __restrict__ src, int C, const float* __restrict__ wd, const float* __restrict__ bd, int act, uch16* Ph, uch16* Pl, int PS) {
        const int cqn = C >> 2, ppl = 256 / cqn;
        const int cq = t % cqn, pl = t / cqn, c = cq * 4;
        if (pl < ppl) {
            float4 w[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) w[k] = *reinterpret_cast<const float4*>(wd + k * C + c);
            const float4 bias = *reinterpret_cast<const float4*>(bd + c);
            for (int r0 = pl; r0 < nrows; r0 += 2 * ppl) {
                float4 win[2][9];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = r0 + u * ppl;
                    const int m = m0 + (r < nrows ? r : r0);
                    const int b = m / HWo, rem = m - b * HWo;
                    const int oy = rem / Wo, ox = rem - oy * Wo;
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky) {
                        const int iy = 2 * oy - 1 + ky;
                        const bool yok = iy >= 0 && iy < a.H;
                        const float* rowp = src + ((size_t)(b * a.H + (yok ? iy : 0)) * a.W) * C + c;
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            const int ix = 2 * ox - 1 + kx;
                            const bool ok = yok && ix >= 0 && ix < a.W;
                            win[u][ky * 3 + kx] = vmask(*reinterpret_cast<const float4*>(rowp + (size_t)(ok ? ix : 0) * C), opaque_mask(ok));
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    const int r = r0 + u * ppl;
                    if (r < nrows) {
                        float4 acc = bias;
#pragma unroll
                        for (int k = 0; k < 9; ++k) vfma(acc, win[u][k], w[k]);
                        acc = vact(acc, act);
                        const float x4[4] = {acc.x, acc.y, acc.z, acc.w};
                        uch16x4 hi, lo;
#pragma unroll
                        for (int j = 0; j < 4; ++j) { amax = range_track(amax, x4[j]); hi[j] = (uch16)x4[j]; lo[j] = (uch16)((x4[j] - (float)hi[j]) * 2048.0f); }
                        *reinterpret_cast<uch16x4*>(Ph + r * PS + c) = hi;
                        *reinterpret_cast<uch16x4*>(Pl + r * PS + c) = lo;
                    }
                }
            }
        }
        // K tail [C, PS) of every row and the idle rows [nrows, BM): zeros (they meet zero weight rows / are never stored, but must not be NaN bit patterns)
        const int padn = PS - C;
        for (int i = t; i < BM * padn; i += 256) { const int r = i / padn, c2 = C + i - r * padn; Ph[r * PS + c2] = (uch16)0.0f; Pl[r * PS + c2] = (uch16)0.0f; }
        for (int i = t; i < (BM - nrows) * (C >> 2); i += 256) {
            const int r = nrows + i / (C >> 2), c2 = (i % (C >> 2)) * 4;
            uch16x4 z; z[0] = z[1] = z[2] = z[3] = (uch16)0.0f;
            *reinterpret_cast<uch16x4*>(Ph + r * PS + c2) = z; *reinterpret_cast<uch16x4*>(Pl + r * PS + c2) = z;
        }
    };
    dw_branch(a.y1, bf, a.wdw, a.bdw, a.dw_act, A2h, A2l, PS2);
    YN_TS();
    dw_branch(a.x, cin, a.wdw1, a.bdw1, a.dw1_act, A1h, A1l, PS1);
    YN_TS();
    // the biases of this lane's columns: requested here (not live during the register-hungry window rounds), used thousands of cycles later
    float bias2[NT], bias3[NT], bias1n[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = (wave * NT + nt) * 32 + l31;
        bias2[nt] = n < bf ? a.b2[n] : 0.0f;
        bias3[nt] = n < bf ? a.b3[n] : 0.0f;
        bias1n[nt] = (chain && n < bf) ? a.b1n[n] : 0.0f;
    }
    __syncthreads();                                                    // both operand tiles are complete
    YN_TS();
    range_report(a.ovf, amax);                                          // every split of this workgroup is done

    // ---- 2. the pointwise convs as ONE walk over their k-steps (gemm_split_tile's order inside each): pw2, branch 1's, then - with a next
    //      unit behind this one - that unit's pw1, whose operand is produced half way (step S): the weight fragments keep streaming across
    //      the two barriers there ---------------------------------------------------------------------------------------------------------
    f32x16 acc0[NT], acc1[NT];
    float y3[NT][16];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc0[i][k] = 0.0f; acc1[i][k] = 0.0f; }
    const int jhi = bf >> 1;
    char* out_base = reinterpret_cast<char*>(a.out + (size_t)m0 * (2 * bf));
    // concat + shuffle store: out[row][2n] = branch 1, out[row][2n + 1] = branch 2; with a next unit behind it, the pairs of the columns
    // n >= bf/2 - channels [bf, 2bf) of the output: that unit's x2 - are ALSO split into the (then free) planes of branch 2
    auto store_pairs = [&]() {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = (wave * NT + nt) * 32 + l31;
            if (n < bf) {
                const bool to_plane = chain && n >= jhi;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
                    const float z = apply_act(__builtin_fmaf(acc1[nt][r], 1.0f / 2048.0f, acc0[nt][r]) + bias3[nt], a.act3);
                    if (row < nrows) *reinterpret_cast<float2*>(out_base + (unsigned)(row * 2 * bf + 2 * n) * 4u) = make_float2(z, y3[nt][r]);
                    if (to_plane) {                                     // (idle rows carry finite values of zero operands: never stored)
                        const float v0 = z, v1 = y3[nt][r];
                        uch16x2 hi, lo;
                        amax = range_track(range_track(amax, v0), v1);
                        hi[0] = (uch16)v0; hi[1] = (uch16)v1;
                        lo[0] = (uch16)((v0 - (float)hi[0]) * 2048.0f); lo[1] = (uch16)((v1 - (float)hi[1]) * 2048.0f);
                        *reinterpret_cast<uch16x2*>(A2h + row * PS2 + 2 * (n - jhi)) = hi;
                        *reinterpret_cast<uch16x2*>(A2l + row * PS2 + 2 * (n - jhi)) = lo;
                    }
                }
            }
        }
    };
    auto run_gemm = [&](const uch16* Ah_, const uch16* Al_, int PS, int Sn, int p0, int p1) {     // padded steps [p0, p1) of one GEMM (Sn real k-steps)
        const uch16* ahp = Ah_ + l31 * PS + h * 8;
        const uch16* alp = Al_ + l31 * PS + h * 8;
        for (int s0 = p0; s0 < p1; s0 += D) {
#pragma unroll
            for (int j = 0; j < D; ++j) {
                const int ks = s0 + j - p0;
                const int a_col = ks < Sn ? ks * 16 : PS - 8 - h * 8;  // padded step: the last 16 bytes of the row's zero K tail
                const uch16x8 ah = *reinterpret_cast<const uch16x8*>(ahp + a_col);
                const uch16x8 al = *reinterpret_cast<const uch16x8*>(alp + a_col);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[j][nt][0], acc0[nt], 0, 0, 0);
                    acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bq[j][nt][1], acc1[nt], 0, 0, 0);
                    acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bq[j][nt][0], acc1[nt], 0, 0, 0);
                }
                load_b(s0 + j + D, bq[j]);                              // (clamped beyond the walk)
            }
        }
    };
    run_gemm(A2h, A2l, PS2, S2, 0, G2);                                 // pw2: its tile then waits in registers for its partner
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            y3[nt][r] = apply_act(__builtin_fmaf(acc1[nt][r], 1.0f / 2048.0f, acc0[nt][r]) + bias2[nt], a.act2);
            acc0[nt][r] = 0.0f; acc1[nt][r] = 0.0f;
        }
    run_gemm(A1h, A1l, PS1, S1, G2, G2 + G1);                           // branch 1's pointwise conv
    if (chain) {                                                        // the unit's output is complete: store it, x2' -> planes, the next unit's pw1
        __syncthreads();                                                // every wavefront is past its last read of the planes
        store_pairs();
        __syncthreads();                                                // x2' is complete (its K tail [bf, PS2) is still zero)
        range_report(a.ovf, amax);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc0[nt][r] = 0.0f; acc1[nt][r] = 0.0f; }
        run_gemm(A2h, A2l, PS2, S2, G2 + G1, G2 + G1 + G2);
    }
    YN_TS();
    if (!chain) {
        store_pairs();
    } else {
        // the next unit's pw1 -> t1n (gemm_split_kernel's epilogue: 16-byte stores through the in-quad transpose)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[nt][r] = __builtin_fmaf(acc1[nt][r], 1.0f / 2048.0f, acc0[nt][r]);
        GemmArgs e{};
        e.out = a.t1n; e.out_ld = bf; e.out_off = 0; e.M = m0 + nrows; e.N = bf; e.Npad = a.Npad; e.bias = a.b1n; e.act = a.act1n; e.pass = nullptr;
        gemm_epilogue<NT>(e, acc0, m0, wave * NT * 32, true, lane, bias1n);
    }
#ifdef YN_EXP_TIMING
    YN_TS();
    if (t == 0 && (blockIdx.x % 97) == 5)
        printf("down2 bf %d blk %d dw2 %lld dw1 %lld sync %lld walk %lld end %lld total %lld\n", bf, (int)blockIdx.x, TS[1] - TS[0], TS[2] - TS[1], TS[3] - TS[2],
               TS[4] - TS[3], TS[5] - TS[4], TS[5] - TS[0]);
#endif
#undef YN_TS
}

static size_t down2_lds(int bf, int cin)
{
    const int PS2 = plane_stride(bf), PS1 = plane_stride(cin);
    return ((size_t)2 * 32 * PS2 + (size_t)2 * 32 * PS1) * 2;
}

bool down2_covers(const Down2Args& a)
{
    return a.x && a.y1 && a.W2h && a.W2l && a.W3h && a.W3l && a.wdw && a.wdw1 && a.bdw && a.bdw1 && a.b2 && a.b3 && a.B > 0 && a.H > 1 && a.W > 1 &&
           a.bf >= 4 && a.cin >= 4 && !(a.bf & 3) && !(a.cin & 3) && a.bf <= 256 && a.cin <= 256 && a.Npad >= a.bf && a.Npad <= 256 &&
           (long)a.B * a.H * a.W * (a.bf > a.cin ? a.bf : a.cin) < (1l << 30);
}

void launch_down2(const Down2Args& a, hipStream_t s)
{
    const int Ho = (a.H - 1) / 2 + 1, Wo = (a.W - 1) / 2 + 1;
    const unsigned tiles = (unsigned)(((long)a.B * Ho * Wo + 31) / 32);
    static unsigned long long attr = 0;
    if (attr_pending(attr)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(down2_kernel<1, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(down2_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (a.Npad <= 128) { set_last_kernel_name("down2_kernel<1,4>"); hipLaunchKernelGGL((down2_kernel<1, 4>), dim3(xcd_grid(tiles)), dim3(256), down2_lds(a.bf, a.cin), s, a); }
    else               { set_last_kernel_name("down2_kernel<2,4>"); hipLaunchKernelGGL((down2_kernel<2, 4>), dim3(xcd_grid(tiles)), dim3(256), down2_lds(a.bf, a.cin), s, a); }
}

// -------------------------------------------------------------------------------------------------
// Depthwise 3x3 + pointwise conv of a detection head (models/yolo_nano.py:60-82: Conv(96, 96, k=3, g=96) -> Conv(96, 96, k=1)) as one
// kernel, for up to three pyramid levels per launch (Group<>).  The depthwise output of the stride-8 head is 33 MB per 32-image step,
// written by one launch and read back by the next; here it goes from registers into the GEMM's LDS operand planes.  Workgroup = an 8 x 4
// tile of pixels of one image: thread = (4 channels, a run of 4 pixels along x) with its 3 x 6 window in ONE batch of clamped, masked
// loads (dwconv3x3_kernel's thread, the same fma chain), the whole 96 x 96 pre-split weight matrix in LDS (no K-chunk barriers), three
// wavefronts = the three 32-column tiles, 16-byte stores through the in-quad transpose.  Bit-identical to dwconv3x3_kernel +
// gemm_split_kernel.  LDS 50 KB: three workgroups per CU.
// -------------------------------------------------------------------------------------------------
template <int TH>                                           // tile height: 8 x TH pixels per workgroup, TH / 4 runs per thread
__device__ __forceinline__ void dwpw_block(const DwPwArgs& a, uch16* smem, unsigned bid, unsigned nblocks)
{
    constexpr int TW = 8, NO = TW * TH, C = 96, KQ = C / 8, BN = 96, AST = C + 8, R = 4, NR = TH / 4;
    uch16* Ah = smem;                                       // [NO][AST]
    uch16* Al = Ah + NO * AST;
    uch16* Bs = Al + NO * AST;                              // [2][KQ][BN][8]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int tx_n = (a.W + TW - 1) / TW, ty_n = (a.H + TH - 1) / TH;
    const int tile = (int)xcd_block(bid, nblocks);
    if (tile >= a.B * ty_n * tx_n) return;
    const int b = tile / (ty_n * tx_n), trem = tile - b * (ty_n * tx_n);
    const int oy0 = (trem / tx_n) * TH, ox0 = (trem % tx_n) * TW;

    // ---- 1. every load of the workgroup in one batch: depthwise windows, taps, bias; the weight matrix; the GEMM bias ------------------
    const int cq = t % (C / 4), run = t / (C / 4);          // 24 channel quads x 8 runs (2 per tile row, rows run/2 + 4 i) = 192 workers
    const bool worker = run < 8;
    const int c = cq * 4, ry = run >> 1, rx = (run & 1) * R;
    float4 win[NR][3][R + 2], wd[9], bd = make_float4(0.f, 0.f, 0.f, 0.f);
    if (worker) {
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int oy = oy0 + ry + 4 * i;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy - 1 + ky;
                const bool yok = oy < a.H && iy >= 0 && iy < a.H;
                const float* rowp = a.in + ((size_t)(b * a.H + (yok ? iy : 0)) * a.W) * C + c;
#pragma unroll
                for (int j = 0; j < R + 2; ++j) {
                    const int ix = ox0 + rx - 1 + j;
                    const bool ok = yok && ix >= 0 && ix < a.W;
                    win[i][ky][j] = vmask(*reinterpret_cast<const float4*>(rowp + (size_t)(ok ? ix : 0) * C), opaque_mask(ok));
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) wd[k] = *reinterpret_cast<const float4*>(a.wdw + k * C + c);
        bd = *reinterpret_cast<const float4*>(a.bdw + c);
    }
    constexpr int B_PER = (2 * KQ * BN + 255) / 256;        // 9 granules of 16 bytes per thread
    uch16x8 b_reg[B_PER];
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int g = t + 256 * i;                          // plane, octet, column
        const int pl = g / (KQ * BN), r = g - pl * (KQ * BN);
        const int o = r / BN, n = r - o * BN;
        uch16x8 v;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (uch16)0.0f;
        if (g < 2 * KQ * BN) v = *reinterpret_cast<const uch16x8*>(reinterpret_cast<const uch16*>(pl ? a.Wl : a.Wh) + ((size_t)o * a.Npad + n) * 8);
        b_reg[i] = v;
    }
    const float gbias = (wave < 3) ? a.bias[wave * 32 + l31] : 0.0f;

    // ---- 2. depthwise (dwconv3x3_kernel's chain) -> split planes; weights -> LDS -------------------------------------------------------
    float amax = 0.0f;                                      // range guard (yn_device.h)
    if (worker) {
#pragma unroll
        for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int o = 0; o < R; ++o) {
            float4 acc = bd;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) vfma(acc, win[i][ky][o + kx], wd[ky * 3 + kx]);
            acc = vact(acc, a.dw_act);
            const int op = (ry + 4 * i) * TW + rx + o;
            const float x4[4] = {acc.x, acc.y, acc.z, acc.w};
            uch16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) { amax = range_track(amax, x4[j]); hi[j] = (uch16)x4[j]; lo[j] = (uch16)((x4[j] - (float)hi[j]) * 2048.0f); }
            *reinterpret_cast<uch16x4*>(Ah + op * AST + c) = hi;
            *reinterpret_cast<uch16x4*>(Al + op * AST + c) = lo;
        }
        range_report(a.ovf, amax);
    }
#pragma unroll
    for (int i = 0; i < B_PER; ++i) {
        const int g = t + 256 * i;
        if (g < 2 * KQ * BN) *reinterpret_cast<uch16x8*>(Bs + (size_t)g * 8) = b_reg[i];
    }
    __syncthreads();

    // ---- 3. pointwise conv: three wavefronts, 32 x 32 each, K = 96 in gemm_split_tile's order; bias, activation, 16-byte stores -----------
    if (wave < 3) {
#pragma unroll
        for (int rt = 0; rt < NO / 32; ++rt) {
        f32x16 acc0, acc1;
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc0[k] = 0.0f; acc1[k] = 0.0f; }
        const uch16* Ahb = Ah + (rt * 32 + l31) * AST + h * 8;
        const uch16* Alb = Al + (rt * 32 + l31) * AST + h * 8;
        const uch16* Bhb = Bs + (size_t)(h * BN + wave * 32 + l31) * 8;
        const uch16* Blb = Bhb + (size_t)KQ * BN * 8;
#pragma unroll
        for (int ks = 0; ks < KQ / 2; ++ks) {
            const uch16x8 ah = *reinterpret_cast<const uch16x8*>(Ahb + ks * 16);
            const uch16x8 al = *reinterpret_cast<const uch16x8*>(Alb + ks * 16);
            const uch16x8 bh = *reinterpret_cast<const uch16x8*>(Bhb + (size_t)(ks * 2 * BN) * 8);
            const uch16x8 bl = *reinterpret_cast<const uch16x8*>(Blb + (size_t)(ks * 2 * BN) * 8);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1, 0, 0, 0);
        }
        const int j = lane & 3;
        const int nq = wave * 32 + (l31 & ~3);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v0 = apply_act(__builtin_fmaf(acc1[4 * g + 0], 1.0f / 2048.0f, acc0[4 * g + 0]) + gbias, a.act);
            float v1 = apply_act(__builtin_fmaf(acc1[4 * g + 1], 1.0f / 2048.0f, acc0[4 * g + 1]) + gbias, a.act);
            float v2 = apply_act(__builtin_fmaf(acc1[4 * g + 2], 1.0f / 2048.0f, acc0[4 * g + 2]) + gbias, a.act);
            float v3 = apply_act(__builtin_fmaf(acc1[4 * g + 3], 1.0f / 2048.0f, acc0[4 * g + 3]) + gbias, a.act);
            {   // 2x2 blocks, then 4x4: lane j of the quad ends up with row j x 4 columns (gemm_epilogue's transpose)
                const float s0 = (j & 1) ? v0 : v1, s1 = (j & 1) ? v2 : v3;
                const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
                if (j & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
            }
            {
                const float s0 = (j & 2) ? v0 : v2, s1 = (j & 2) ? v1 : v3;
                const float r0 = quad_xor2(s0), r1 = quad_xor2(s1);
                if (j & 2) { v0 = r0; v1 = r1; } else { v2 = r0; v3 = r1; }
            }
            const int op = rt * 32 + 8 * g + 4 * h + j;
            const int py = oy0 + op / TW, px = ox0 + op % TW;
            if (py < a.H && px < a.W)
                *reinterpret_cast<float4*>(a.out + (((size_t)b * a.H + py) * a.W + px) * C + nq) = make_float4(v0, v1, v2, v3);
        }
        }
    }
}

template <int TH>
__global__ __launch_bounds__(256, (TH == 4 ? 3 : 2)) void dwpw_group_kernel(Group<DwPwArgs> g)
{
    extern __shared__ __attribute__((aligned(16))) float dwpw_smem[];
    unsigned local, nb;
    const int p = group_problem(g.first, blockIdx.x, local, nb);
    dwpw_block<TH>(g.a[p], reinterpret_cast<uch16*>(dwpw_smem), local, nb);
}

// -------------------------------------------------------------------------------------------------
// dwpw_block as a tile walk (round 4, after down_unit_pipe_kernel): 192 threads (the fourth wavefront of dwpw_block only helped staging
// the weight matrix), the 96 x 96 split weights register-resident (a lane's B fragments of its wavefront's 32 columns: 48 registers, loaded
// once per workgroup - nothing of them in LDS: 50 -> 13 KB), a workgroup walks ~3 tiles (XCD-contiguous) and requests the next tile's
// depthwise windows right behind the barrier that ends the depthwise phase - the round trip runs under the GEMM and the stores.  Same
// arithmetic in the same order: bit-identical (test_dwpw_fused_is_bit_identical).
// -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void dwpw_pipe_block(const DwPwArgs& a, uch16* smem, unsigned bid, unsigned nblocks)
{
    constexpr int TW = 8, TH = 4, NO = TW * TH, C = 96, KQ = C / 8, KS = KQ / 2, AST = C + 8, R = 4;
    uch16* Ah = smem;                                       // [NO][AST]
    uch16* Al = Ah + NO * AST;
    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, h = lane >> 5;
    const int tx_n = (a.W + TW - 1) / TW, ty_n = (a.H + TH - 1) / TH, per_img = tx_n * ty_n;
    const int tiles = a.B * per_img;
    const int TL = (tiles + 7) >> 3, GL = (int)(nblocks >> 3);
    const int tbase = (int)(bid & 7u) * TL;
    int tl = (int)(bid >> 3);
    if (tl >= TL || tbase + tl >= tiles) return;

    // ---- loop invariants: the thread's depthwise taps and bias, the B fragments and bias of its GEMM columns --------------------------
    const int cq = t % (C / 4), run = t / (C / 4);          // 24 channel quads x 8 runs of 4 pixels (2 per tile row) = 192 workers
    const int c = cq * 4, ry = run >> 1, rx = (run & 1) * R;
    float4 wd[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) wd[k] = *reinterpret_cast<const float4*>(a.wdw + k * C + c);
    const float4 bd = *reinterpret_cast<const float4*>(a.bdw + c);
    uch16x8 gh[KS], gl[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const size_t off = ((size_t)(ks * 2 + h) * a.Npad + wave * 32 + l31) * 8;
        gh[ks] = *reinterpret_cast<const uch16x8*>(reinterpret_cast<const uch16*>(a.Wh) + off);
        gl[ks] = *reinterpret_cast<const uch16x8*>(reinterpret_cast<const uch16*>(a.Wl) + off);
    }
    const float gbias = a.bias[wave * 32 + l31];
    float4 win[3][R + 2];
    unsigned wok = 0;                                       // validity of the 18 window positions of the pending request
    const char* xbase = reinterpret_cast<const char*>(a.in);
    auto request = [&](int tile) {
        const int b = tile / per_img, trem = tile - b * per_img;
        const int oy = (trem / tx_n) * TH + ry, ox = (trem % tx_n) * TW + rx;
        wok = 0;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy - 1 + ky;
            const bool yok = oy < a.H && iy >= 0 && iy < a.H;
#pragma unroll
            for (int j = 0; j < R + 2; ++j) {
                const int ix = ox - 1 + j;
                const bool ok = yok && ix >= 0 && ix < a.W;
                // raw value from a clamped address, zeroed where it is consumed (a mask applied here would wait for the load here)
                win[ky][j] = *reinterpret_cast<const float4*>(xbase + (unsigned)(ok ? ((b * a.H + iy) * a.W + ix) * C + c : c) * 4u);
                wok |= (ok ? 1u : 0u) << (ky * (R + 2) + j);
            }
        }
    };
    request(tbase + tl);
    float amax = 0.0f;                                      // range guard (yn_device.h)
    const int j4 = lane & 3;
    const int nq = wave * 32 + (l31 & ~3);

    for (;;) {
        const int tile = tbase + tl;
        const int b = tile / per_img, trem = tile - b * per_img;
        const int oy0 = (trem / tx_n) * TH, ox0 = (trem % tx_n) * TW;
        // ---- 1. depthwise (dwconv3x3_kernel's chain) -> split planes ---------------------------------------------------------------
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int j = 0; j < R + 2; ++j) win[ky][j] = vmask(win[ky][j], 0u - ((wok >> (ky * (R + 2) + j)) & 1u));
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
            uch16x4 hi, lo;
#pragma unroll
            for (int j = 0; j < 4; ++j) { amax = range_track(amax, x4[j]); hi[j] = (uch16)x4[j]; lo[j] = (uch16)((x4[j] - (float)hi[j]) * 2048.0f); }
            *reinterpret_cast<uch16x4*>(Ah + op * AST + c) = hi;
            *reinterpret_cast<uch16x4*>(Al + op * AST + c) = lo;
        }
        __syncthreads();
        const bool more = tl + GL < TL && tbase + tl + GL < tiles;
        if (more) request(tbase + tl + GL);                 // the next tile's windows fly under the GEMM and the stores

        // ---- 2. pointwise conv: 32 x 32 per wavefront, K = 96 in gemm_split_tile's order; bias, activation, 16-byte stores -------------
        f32x16 acc0, acc1;
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc0[k] = 0.0f; acc1[k] = 0.0f; }
        const uch16* Ahb = Ah + l31 * AST + h * 8;
        const uch16* Alb = Al + l31 * AST + h * 8;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const uch16x8 ah = *reinterpret_cast<const uch16x8*>(Ahb + ks * 16);
            const uch16x8 al = *reinterpret_cast<const uch16x8*>(Alb + ks * 16);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gh[ks], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, gl[ks], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, gh[ks], acc1, 0, 0, 0);
        }
        // tile pixel of accumulator group g, lane (h, j4): row g, column 4 h + j4 - a wave-uniform row offset + one lane offset, 32-bit
        char* obase = reinterpret_cast<char*>(a.out);
        const unsigned lane_off = (unsigned)(((b * a.H + oy0) * a.W + ox0 + 4 * h + j4) * C + nq) * 4u;
        const bool xin = ox0 + 4 * h + j4 < a.W;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v0 = apply_act(__builtin_fmaf(acc1[4 * g + 0], 1.0f / 2048.0f, acc0[4 * g + 0]) + gbias, a.act);
            float v1 = apply_act(__builtin_fmaf(acc1[4 * g + 1], 1.0f / 2048.0f, acc0[4 * g + 1]) + gbias, a.act);
            float v2 = apply_act(__builtin_fmaf(acc1[4 * g + 2], 1.0f / 2048.0f, acc0[4 * g + 2]) + gbias, a.act);
            float v3 = apply_act(__builtin_fmaf(acc1[4 * g + 3], 1.0f / 2048.0f, acc0[4 * g + 3]) + gbias, a.act);
            {   // 2x2 blocks, then 4x4: lane j of the quad ends up with row j x 4 columns (gemm_epilogue's transpose)
                const float s0 = (j4 & 1) ? v0 : v1, s1 = (j4 & 1) ? v2 : v3;
                const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
                if (j4 & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
            }
            {
                const float s0 = (j4 & 2) ? v0 : v2, s1 = (j4 & 2) ? v1 : v3;
                const float r0 = quad_xor2(s0), r1 = quad_xor2(s1);
                if (j4 & 2) { v0 = r0; v1 = r1; } else { v2 = r0; v3 = r1; }
            }
            if (oy0 + g < a.H && xin)
                *reinterpret_cast<float4*>(obase + (lane_off + (unsigned)(g * a.W * C) * 4u)) = make_float4(v0, v1, v2, v3);
        }
        if (!more) break;
        tl += GL;
        __syncthreads();                                    // every wavefront is done with the planes
    }
    range_report(a.ovf, amax);
}

__global__ __launch_bounds__(192, 2) void dwpw_pipe_group_kernel(Group<DwPwArgs> g)
{
    extern __shared__ __attribute__((aligned(16))) float dwpw_smem[];
    unsigned local, nb;
    const int p = group_problem(g.first, blockIdx.x, local, nb);
    dwpw_pipe_block(g.a[p], reinterpret_cast<uch16*>(dwpw_smem), local, nb);
}

bool dwpw_group_ok(const DwPwArgs* a, int n)
{
    if (n < 1 || n > YN_GROUP_MAX) return false;
    for (int p = 0; p < n; ++p)
        if (a[p].C != 96 || a[p].Npad != 96 || !a[p].Wh || !a[p].Wl || a[p].B <= 0) return false;
    return true;
}

void launch_dwpw_group(const DwPwArgs* a, int n, hipStream_t s)
{
    static const int th_env = getenv("YN_DWPW_TH") ? atoi(getenv("YN_DWPW_TH")) : 0;
    unsigned tiles4 = 0;
    for (int p = 0; p < n; ++p) tiles4 += (unsigned)a[p].B * ((a[p].H + 3) / 4) * ((a[p].W + 7) / 8);
    (void)tiles4;
    const int TH = th_env == 8 ? 8 : 4;                     // 64-pixel tiles (half the weight traffic) measured the same end to end, 43 vs 39 us alone: YN_DWPW_TH=8 keeps them for A/B runs
    Group<DwPwArgs> g{};
    unsigned tot = 0;
    // the tile-walking form (8 x 4 tiles only; YN_DWPW_PIPE=0: one tile per workgroup; YN_DWPW_PIPE_T: tiles per walking workgroup)
    static const int pipe = getenv("YN_DWPW_PIPE") ? atoi(getenv("YN_DWPW_PIPE")) : 1;
    static const int pipe_t = getenv("YN_DWPW_PIPE_T") ? atoi(getenv("YN_DWPW_PIPE_T")) : 3;
    if (pipe && TH == 4) {
        bool fits = true;
        for (int p = 0; p < n; ++p) fits = fits && (size_t)a[p].B * a[p].H * a[p].W * 96 < ((size_t)1 << 30);      // 32-bit byte offsets
        if (fits) {
            const unsigned per = (unsigned)(pipe_t > 0 ? pipe_t : 1);
            for (int p = 0; p < YN_GROUP_MAX; ++p) {
                g.first[p] = tot;
                if (p < n) { g.a[p] = a[p]; const unsigned tiles = (unsigned)a[p].B * ((a[p].H + 3) / 4) * ((a[p].W + 7) / 8); tot += xcd_grid((tiles + per - 1) / per); }
            }
            g.first[YN_GROUP_MAX] = tot;
            set_last_kernel_name("dwpw_pipe_group_kernel");
            hipLaunchKernelGGL(dwpw_pipe_group_kernel, dim3(tot), dim3(192), (size_t)2 * 32 * 104 * 2, s, g);
            return;
        }
    }
    for (int p = 0; p < YN_GROUP_MAX; ++p) {
        g.first[p] = tot;
        if (p < n) { g.a[p] = a[p]; tot += xcd_grid((unsigned)a[p].B * ((a[p].H + TH - 1) / TH) * ((a[p].W + 7) / 8)); }
    }
    g.first[YN_GROUP_MAX] = tot;
    const size_t lds = ((size_t)2 * 8 * TH * 104 + (size_t)2 * 12 * 96 * 8) * 2;
    static unsigned long long attr = 0;
    if (attr_pending(attr)) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwpw_group_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dwpw_group_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    }
    if (TH == 8) { set_last_kernel_name("dwpw_group_kernel<8>"); hipLaunchKernelGGL(dwpw_group_kernel<8>, dim3(tot), dim3(256), lds, s, g); }
    else         { set_last_kernel_name("dwpw_group_kernel<4>"); hipLaunchKernelGGL(dwpw_group_kernel<4>, dim3(tot), dim3(256), lds, s, g); }
}

bool launch_unit_chain(const ChainArgs& a, hipStream_t s) { return unit_chain_dispatch(a, s, false); }
bool unit_chain_covers(const ChainArgs& a) { return unit_chain_dispatch(a, nullptr, true); }

}  // namespace ynk

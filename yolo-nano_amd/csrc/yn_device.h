// yn_device.h — device-side helpers shared by the kernel translation units (kernels_conv.hip, kernels_chain.hip):
// activation, the opaque-mask load idiom, the GEMM epilogue of the f32 MFMA accumulator layout, the split-f16 GEMM tile, small vector helpers.
#pragma once
#include "yn_internal.h"

namespace ynk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ---- range guard of the split-f16 family ----------------------------------------------------------------------------------
// x = hi + lo * 2^-11 takes hi = (f16)x: finite only for |x| < 65520.  Beyond that hi = +-inf, lo = -+inf, the three-MFMA sum is
// NaN (and a ReLU epilogue turns that NaN into 0), where the reference's fp32 conv is finite.  Every kernel that splits
// activations keeps the running max |x| of what it splits (one v_max_f32 per element, next to the five VALU ops of the split itself)
// and raises the handle's flag once per wavefront at its end; yn_range_status() reports it and the host shim re-runs on the f32-MFMA
// family (yn_exact_f32).  Folded WEIGHTS are checked once, at yn_fold_bn (fold_pack_kernel).  Tiny values need no guard: below the f16
// normal range hi loses bits (or flushes to 0) but lo = (x - hi) * 2^11 still carries x exactly to 11 bits more, i.e. an absolute
// error <= 2^-25 * 2^-11 - far below the fp32 round-off of any accumulation that also holds O(1) terms.
#ifdef YN_EXP_NO_RANGE                                      // timing experiment only: the guard compiled out
__device__ __forceinline__ float range_track(float amax, float) { return amax; }
__device__ __forceinline__ void range_report(unsigned*, float) {}
#else
__device__ __forceinline__ float range_track(float amax, float x) { return __builtin_fmaxf(amax, __builtin_fabsf(x)); }
__device__ __forceinline__ void range_report(unsigned* ovf, float amax)
{
    if (ovf && amax >= 65504.0f) atomicOr(ovf, 1u);         // +inf included; a NaN input is NaN in the reference too
}
#endif

// Activation without control flow: with a run-time `act` an if-chain compiles to branches PER VALUE in the unrolled epilogues (three
// per accumulator register, ~250 in one pointwise-GEMM kernel).  x > 0 ? x : (act 1: +0, act 2: 0.1 x, act 0: 1.0 x = x); NaN takes
// the second operand: the same bits as the if-chain this replaces.
__device__ __forceinline__ float apply_act(float v, int act)
{
    const float slope = act == 2 ? 0.1f : 1.0f;            // wave-uniform: two scalar selects, hoisted out of the epilogue loops
    const unsigned keep = act == 1 ? 0u : 0xffffffffu;
    const float neg = __uint_as_float(__float_as_uint(slope * v) & keep);
    return v > 0.0f ? v : neg;
}

// lane ^ 1 / lane ^ 2 exchanges inside a quad as DPP quad_perm moves (one VALU instruction; __shfl_xor compiles to ds_bpermute_b32, an
// LDS round trip: 16 of them per 32 output columns in the 16-byte-store epilogues)
__device__ __forceinline__ float quad_xor1(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xf, 0xf, false)); }   // quad_perm:[1,0,3,2]
__device__ __forceinline__ float quad_xor2(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xf, 0xf, false)); }   // quad_perm:[2,3,0,1]

// Loads whose result is only sometimes wanted are issued UNCONDITIONALLY at a clamped (always legal) address and the
// unwanted values are zeroed with a bit mask the optimiser cannot see through.  `if (ok) v = load` — and `ok ? load : 0`,
// and `load & mask` with a visible mask — all compile to a branch around the load followed by s_waitcnt vmcnt(0), i.e.
// one full memory latency per load instead of one per batch of loads.
__device__ __forceinline__ unsigned opaque_mask(bool ok)
{
    unsigned mk = ok ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(mk));
    return mk;
}
__device__ __forceinline__ float2 vmask(float2 v, unsigned mk)
{
    return make_float2(__uint_as_float(__float_as_uint(v.x) & mk), __uint_as_float(__float_as_uint(v.y) & mk));
}
__device__ __forceinline__ float4 vmask(float4 v, unsigned mk)
{
    return make_float4(__uint_as_float(__float_as_uint(v.x) & mk), __uint_as_float(__float_as_uint(v.y) & mk),
                       __uint_as_float(__float_as_uint(v.z) & mk), __uint_as_float(__float_as_uint(v.w) & mk));
}

// ---- GEMM epilogue shared by the tiled and the persistent kernel: bias + activation (+ concat/shuffle interleave with
//      the pass-through half).  mbase / nbase = first row / column of this wave's 32 x (32*NT) accumulator block.
template <int NT, bool PASS>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmArgs& a, f32x16 (&acc)[NT], int mbase, int nbase, bool vecO, int lane,
                                              const float* pre_bias = nullptr)   // pre_bias[nt]: bias of this lane's column, loaded earlier
{
    const int l31 = lane & 31, h = lane >> 5;
    if (vecO) {
        // 16-byte stores: an accumulator quad (regs 4g..4g+3 = 4 consecutive rows, lanes 4q'..4q'+3 = 4 consecutive
        // columns) is transposed inside its 4 lanes with two xor-shuffles, so lane j ends up with row j x 4 columns.
        const int j = lane & 3;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int ncol = nbase + nt * 32 + l31;            // this lane's column before the transpose
            const float bias = pre_bias ? pre_bias[nt] : (ncol < a.N ? a.bias[ncol] : 0.0f);
            const int nq = nbase + nt * 32 + (l31 & ~3);       // first column of the quad
            // the pass-through half of the four row groups: requested together, before the transposes (issued one by one
            // inside the `if (m < M)` below, each load is followed by a full wait)
            float4 pv[4];
            if (PASS) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int m = mbase + 8 * g + 4 * h + j;
                    const bool ok = m < a.M && nq < a.N;
                    pv[g] = *reinterpret_cast<const float4*>(a.pass + (size_t)(ok ? m : 0) * a.pass_ld + a.pass_off + (ok ? nq : 0));
                }
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = apply_act(acc[nt][4 * g + 0] + bias, a.act), v1 = apply_act(acc[nt][4 * g + 1] + bias, a.act);
                float v2 = apply_act(acc[nt][4 * g + 2] + bias, a.act), v3 = apply_act(acc[nt][4 * g + 3] + bias, a.act);
                {   // 2x2 blocks
                    const float s0 = (j & 1) ? v0 : v1, s1 = (j & 1) ? v2 : v3;
                    const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
                    if (j & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
                }
                {   // 4x4
                    const float s0 = (j & 2) ? v0 : v2, s1 = (j & 2) ? v1 : v3;
                    const float r0 = quad_xor2(s0), r1 = quad_xor2(s1);
                    if (j & 2) { v0 = r0; v1 = r1; } else { v2 = r0; v3 = r1; }
                }
                const int m = mbase + 8 * g + 4 * h + j;
                if (m < a.M && nq < a.N) {
                    if (PASS) {
                        const float4 p = pv[g];
                        float* o = a.out + (size_t)m * a.out_ld + a.out_off + 2 * nq;
                        *reinterpret_cast<float4*>(o) = make_float4(p.x, v0, p.y, v1);
                        *reinterpret_cast<float4*>(o + 4) = make_float4(p.z, v2, p.w, v3);
                    } else {
                        *reinterpret_cast<float4*>(a.out + (size_t)m * a.out_ld + a.out_off + nq) = make_float4(v0, v1, v2, v3);
                    }
                }
            }
        }
        return;
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = nbase + nt * 32 + l31;
        if (n >= a.N) continue;
        const float bias = pre_bias ? pre_bias[nt] : a.bias[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = mbase + row;
            if (m >= a.M) continue;
            const float v = apply_act(acc[nt][r] + bias, a.act);
            if (PASS) {
                const float p = a.pass[(size_t)m * a.pass_ld + a.pass_off + n];
                *reinterpret_cast<float2*>(a.out + (size_t)m * a.out_ld + a.out_off + 2 * n) = make_float2(p, v);
            } else {
                a.out[(size_t)m * a.out_ld + a.out_off + n] = v;
            }
        }
    }
}


// the pass-through variant is selected ONCE: inside the unrolled loops a run-time `a.pass` test is a branch per accumulator group
template <int NT>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x16 (&acc)[NT], int mbase, int nbase, bool vecO, int lane,
                                              const float* pre_bias = nullptr)
{
    if (a.pass) gemm_epilogue_impl<NT, true>(a, acc, mbase, nbase, vecO, lane, pre_bias);
    else gemm_epilogue_impl<NT, false>(a, acc, mbase, nbase, vecO, lane, pre_bias);
}

// ---- split-f16 pointwise GEMM tile (gemm_split_kernel and head_decode_kernel share it): the block's 32*WM x 32*NT*WN tile of
//      in[M][K] x W[K][Npad] as three f16 MFMAs per 16-deep k-step on split fp32 operands (x = hi + lo*2^-11), K in
//      chunks of 32 through LDS with the next chunk's global loads in flight during the MFMAs.  acc0 returns the wave's
//      32 x (32*NT) block (rows m0 + 32*wm.., columns n0 + 32*NT*wn..) BEFORE bias; the MFMA sequence per accumulator only
//      depends on K, so every caller / tile shape gives bit-identical sums.  Ends with all waves past their last LDS read
//      only after the caller's next __syncthreads().
typedef _Float16 c3h16;
typedef _Float16 c3h16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 c3h16x4 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int gemm_split_smem_halves(int BM, int BN, int KC = 32) { return 2 * BM * (KC + 8) + 2 * (KC / 8) * BN * 8; }

// KC = K per chunk (32, or 64: half the barrier rounds for the small-M / long-K layers; zero-padded chunks add exact zeros, same bits)
template <int WM, int WN, int NT, int KC = 32>
__device__ __forceinline__ void gemm_split_tile(const GemmArgs& a, c3h16* smem, int m0, int n0, f32x16 (&acc0)[NT])
{
    constexpr int BM = 32 * WM, BN = 32 * NT * WN, AST = KC + 8, OQ = KC / 8;
    constexpr int A_PER = (BM * OQ + 255) / 256;            // (row, octet) granules per thread per chunk
    constexpr int B_PER = (2 * OQ * BN + 255) / 256;        // 16-byte granules per thread per chunk (hi and lo planes)
    c3h16* Ah = smem;
    c3h16* Al = smem + BM * AST;
    c3h16* Bh = smem + 2 * BM * AST;                        // [OQ][BN][8], then the lo plane
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    const int KQ = (a.K + 7) >> 3, nchunks = (a.K + KC - 1) / KC;
    const bool vecA = ((a.K | a.in_ld | a.in_off) & 3) == 0;
    const c3h16* Wsh = reinterpret_cast<const c3h16*>(a.Wsh);
    const c3h16* Wsl = reinterpret_cast<const c3h16*>(a.Wsl);

    float4 a_reg[A_PER][2];
    c3h16x8 b_reg[B_PER];
    float amax = 0.0f;                                      // range guard: largest |activation| this thread has split
    // prefetch(): NOTHING but loads (round 5).  The round-2 form masked every loaded value where it was loaded (vmask on A, a select on B): a
    // use at the point of issue, so hipcc waited for each load in turn - vmcnt(3) ... vmcnt(0) behind every group of four - BEFORE the MFMAs
    // of the current chunk the prefetch was meant to hide under; the K loop ran load -> wait -> MFMA -> barrier, strictly in sequence.  Now the
    // addresses are clamped (row M - 1, column Npad - 1, octet KQ - 1, the row start for k >= K) and the only thing that must be zero - the A
    // values past K, which meet the clamped octets - is zeroed in stage(), after the MFMAs, where the values are needed anyway.  Rows past M and
    // columns past Npad compute on duplicates and are never stored.  Same values in the planes: the same bits.
    auto prefetch = [&](int c) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int g = t + 256 * i;
            const int row = g / OQ, k = c * KC + (g % OQ) * 8;
            const int m = m0 + row;
            const float* p = a.in + (size_t)(m < a.M ? m : a.M - 1) * a.in_ld + a.in_off;
            if (vecA) {
                a_reg[i][0] = *reinterpret_cast<const float4*>(p + (k < a.K ? k : 0));
                a_reg[i][1] = *reinterpret_cast<const float4*>(p + (k + 4 < a.K ? k + 4 : 0));
            } else {
                float2 v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = *reinterpret_cast<const float2*>(p + (k + 2 * j < a.K ? k + 2 * j : 0));
                a_reg[i][0] = make_float4(v[0].x, v[0].y, v[1].x, v[1].y);
                a_reg[i][1] = make_float4(v[2].x, v[2].y, v[3].x, v[3].y);
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int g = t + 256 * i;                      // plane, octet, column
            const int pl = g / (OQ * BN), r = g - pl * (OQ * BN);
            const int o = r / BN, n = r - o * BN;
            const int kq = min(c * (KC / 8) + o, KQ - 1), nn = min(n0 + n, a.Npad - 1);
            b_reg[i] = *reinterpret_cast<const c3h16x8*>(((pl & 1) ? Wsl : Wsh) + ((size_t)kq * a.Npad + nn) * 8);
        }
    };
    auto stage = [&](int c) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int g = t + 256 * i;
            if (g < BM * OQ) {
                const int k = c * KC + (g % OQ) * 8;        // valid values of this granule: K - k of them (vecA: whole quads, else pairs)
                float x8[8] = {a_reg[i][0].x, a_reg[i][0].y, a_reg[i][0].z, a_reg[i][0].w, a_reg[i][1].x, a_reg[i][1].y, a_reg[i][1].z, a_reg[i][1].w};
#pragma unroll
                for (int j = 0; j < 8; ++j) x8[j] = (k + (vecA ? (j & 4) : (j & 6)) < a.K) ? x8[j] : 0.0f;
                c3h16x8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) { amax = range_track(amax, x8[j]); hi[j] = (c3h16)x8[j]; lo[j] = (c3h16)((x8[j] - (float)hi[j]) * 2048.0f); }
                *reinterpret_cast<c3h16x8*>(Ah + (g / OQ) * AST + (g % OQ) * 8) = hi;
                *reinterpret_cast<c3h16x8*>(Al + (g / OQ) * AST + (g % OQ) * 8) = lo;
            }
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int g = t + 256 * i;
            if (g < 2 * OQ * BN) *reinterpret_cast<c3h16x8*>(Bh + (size_t)g * 8) = b_reg[i];
        }
    };
    f32x16 acc1[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[i][r] = 0.0f; acc1[i][r] = 0.0f; }

    prefetch(0);
    stage(0);
    __syncthreads();
    for (int c = 0; c < nchunks; ++c) {
        if (c + 1 < nchunks) prefetch(c + 1);
        const c3h16* Ahb = Ah + (wm * 32 + l31) * AST + h * 8;
        const c3h16* Alb = Al + (wm * 32 + l31) * AST + h * 8;
        const c3h16* Bhb = Bh + (size_t)(h * BN + wn * NT * 32 + l31) * 8;
        const c3h16* Blb = Bhb + OQ * BN * 8;
#pragma unroll
        for (int ks = 0; ks < KC / 16; ++ks) {
            const c3h16x8 ah = *reinterpret_cast<const c3h16x8*>(Ahb + ks * 16);
            const c3h16x8 al = *reinterpret_cast<const c3h16x8*>(Alb + ks * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const c3h16x8 bh = *reinterpret_cast<const c3h16x8*>(Bhb + (size_t)(ks * 2 * BN + nt * 32) * 8);
                const c3h16x8 bl = *reinterpret_cast<const c3h16x8*>(Blb + (size_t)(ks * 2 * BN + nt * 32) * 8);
                acc0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0[nt], 0, 0, 0);
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1[nt], 0, 0, 0);
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1[nt], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) {
            __syncthreads();
            stage(c + 1);
            __syncthreads();
        }
    }
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[nt][r] = __builtin_fmaf(acc1[nt][r], 1.0f / 2048.0f, acc0[nt][r]);
    range_report(a.ovf, amax);
}

// Row stride (halves) of an operand plane [rows][C] in LDS that 16-byte fragment reads walk row by row (lane = row): a
// ds_read_b128 is served in groups of 16 lanes, conflict-free when their 16-byte pieces tile the 64 banks, i.e. when the stride is an
// ODD multiple of 16 bytes.  ceil(C/8)*8 + 8 is one only for an even octet count: C = 116 (15 octets) gave 256 bytes - all 16 lanes
// on the same four banks (SQ_LDS_BANK_CONFLICT = 88 % of the stage-3 chain's LDS cycles, profiles/r04_sq_counters.txt) - and C = 232 a
// two-way conflict.  The columns [C, stride) stay zero (K tail).
__host__ __device__ constexpr int plane_stride(int C) { return ((((C + 7) >> 3) + 1) & ~1) * 8 + 8; }

// ---- LDS-DMA (global_load_lds_dwordx4) and the barriers that go with it (unit_pipe_kernel, head_tail_pipe_group_kernel) -----------------
// One LDS-DMA piece: 64 lanes x 16 bytes, global (wave-uniform base + 32-bit lane offset) -> LDS (wave-uniform byte address + lane * 16).
// Invisible to hipcc's s_waitcnt bookkeeping (cdna_hip_programming.md 5.x "What hipcc does not do"): completion is counted by hand below.
__device__ __forceinline__ void dma16(const void* gbase, unsigned goff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(goff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}
// barriers that do NOT drain the vector-memory counter (a __syncthreads() may: its fence waits for this wavefront's global stores, and the
// in-order counter then retires the DMA pieces in front of them too)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
__device__ __forceinline__ void vm_drain() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// Raw-buffer stores: SGPR descriptor + ONE 32-bit lane offset + a 12-bit immediate - the four rows of an accumulator group share an offset register
// where `global_store` held a 64-bit lane address per row as a loop invariant (unit_pipe_kernel<116,false,8> spilled them: 16 bytes of scratch whose
// reload - s_waitcnt vmcnt(0) - retired the DMA pieces in flight).  AUX = 0: plain; 16: sc1 = write-through (hand-off to another workgroup inside a
// launch, kernels_stage.hip).  Offsets are unsigned 32-bit (the descriptor spans 4 GB).
__device__ __forceinline__ __amdgpu_buffer_rsrc_t buf_rsrc(void* p) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, -1, 0x00020000); }
template <int AUX>
__device__ __forceinline__ void buf_store_b128(__amdgpu_buffer_rsrc_t r, unsigned off, float4 v)
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    typedef unsigned u4 __attribute__((__vector_size__(4 * sizeof(unsigned))));
    const f4 d = {v.x, v.y, v.z, v.w};
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4, d), r, (int)off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void buf_store_b64(__amdgpu_buffer_rsrc_t r, unsigned off, float2 v)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef unsigned u2 __attribute__((__vector_size__(2 * sizeof(unsigned))));
    const f2 d = {v.x, v.y};
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u2, d), r, (int)off, 0, AUX);
}

template <int V> struct VecT;
template <> struct VecT<2> { typedef float2 type; };
template <> struct VecT<4> { typedef float4 type; };
// explicit fused multiply-adds: with `acc += v * w` the compiler is free to contract or not per instance (it split the first
// tap into v_pk_mul + v_add in one kernel and fused it in another), and the depthwise conv must round identically wherever
// it is evaluated (dwconv3x3_kernel, unit_chain_kernel)
__device__ __forceinline__ void vfma(float2& acc, const float2 v, const float2 w) { acc.x = __builtin_fmaf(v.x, w.x, acc.x); acc.y = __builtin_fmaf(v.y, w.y, acc.y); }
__device__ __forceinline__ void vfma(float4& acc, const float4 v, const float4 w)
{
    acc.x = __builtin_fmaf(v.x, w.x, acc.x); acc.y = __builtin_fmaf(v.y, w.y, acc.y);
    acc.z = __builtin_fmaf(v.z, w.z, acc.z); acc.w = __builtin_fmaf(v.w, w.w, acc.w);
}
__device__ __forceinline__ float2 vact(float2 v, int act) { return make_float2(apply_act(v.x, act), apply_act(v.y, act)); }
__device__ __forceinline__ float4 vact(float4 v, int act) { return make_float4(apply_act(v.x, act), apply_act(v.y, act), apply_act(v.z, act), apply_act(v.w, act)); }

}  // namespace ynk

// yn_device.h — device-side helpers shared by the kernel translation units (kernels_conv.hip, kernels_chain.hip):
// activation, the opaque-mask load idiom, the GEMM epilogue of the f32 MFMA accumulator layout, small vector helpers.
#pragma once
#include "yn_internal.h"

namespace ynk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float apply_act(float v, int act)
{
    if (act == 1) return v > 0.0f ? v : 0.0f;
    if (act == 2) return v > 0.0f ? v : 0.1f * v;
    return v;
}

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
template <int NT>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& a, f32x16 (&acc)[NT], int mbase, int nbase, bool vecO, int lane,
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
            if (a.pass) {
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
                    const float r0 = __shfl_xor(s0, 1), r1 = __shfl_xor(s1, 1);
                    if (j & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
                }
                {   // 4x4
                    const float s0 = (j & 2) ? v0 : v2, s1 = (j & 2) ? v1 : v3;
                    const float r0 = __shfl_xor(s0, 2), r1 = __shfl_xor(s1, 2);
                    if (j & 2) { v0 = r0; v1 = r1; } else { v2 = r0; v3 = r1; }
                }
                const int m = mbase + 8 * g + 4 * h + j;
                if (m < a.M && nq < a.N) {
                    if (a.pass) {
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
            if (a.pass) {
                const float p = a.pass[(size_t)m * a.pass_ld + a.pass_off + n];
                *reinterpret_cast<float2*>(a.out + (size_t)m * a.out_ld + a.out_off + 2 * n) = make_float2(p, v);
            } else {
                a.out[(size_t)m * a.out_ld + a.out_off + n] = v;
            }
        }
    }
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

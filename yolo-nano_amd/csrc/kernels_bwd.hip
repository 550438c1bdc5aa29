// kernels_bwd.hip — train-mode kernels of the YOLO-Nano hot path: BatchNorm with batch statistics
// (forward + backward), weight gradients of the pointwise / dense 3x3 / depthwise / stem convolutions, input gradients
// that cannot reuse a forward kernel (stride-2 depthwise, max pool), and the element-wise glue of the FPN/PAN adds and
// the concat+shuffle.  Input gradients of the pointwise, dense-3x3 and stride-1 depthwise convolutions reuse the forward
// kernels of kernels_conv.hip on transposed / flipped packed weights (pack_bwd_kernel).
//
// The reference has no code for any of this beyond torch autograd (train.py:219-231); semantics are those of
// nn.BatchNorm2d(momentum=0.1, eps=1e-5).train(), F.conv2d / F.max_pool2d / F.interpolate(nearest) backward.
// Correctness-first round-1 implementations (fp32, NHWC); weight gradients accumulate with float atomics.
#include "yn_internal.h"
#include <stdlib.h>

namespace ynk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// =================================================================================================
// Column reductions over an [M, C] matrix and the BatchNorm kernels built on them.
//
// Thread layout of every kernel here: a lane owns a PAIR of adjacent channels (one float2 per row) and walks down the
// rows, `lanesC` (a power of two >= C/2, at most 256) lanes side by side and 256/lanesC rows per block iteration, so a
// wave reads whole consecutive NHWC rows.  Reductions accumulate in double, combine the row-lanes of a block through LDS
// and add ONE double atomic per channel per block into a zeroed accumulator; the consumer kernels turn the sums into
// mean / invstd / gradients themselves, so there is no finalize launch.
//   stats   : acc[0][c] = sum y,   acc[1][c] = sum y*y                  (one pass; E[y^2]-E[y]^2 in double)
//   bwd sums: acc[0][c] = sum dyh, acc[1][c] = sum dyh * xhat           (dyh = dz * act'(z), xhat = (y-mean)*invstd)
// =================================================================================================
// Every streaming loop below loads a BATCH of rows with unconditional (clamped-address) loads before it touches any of
// them: these kernels are latency-bound otherwise (one dependent 8-byte load per lane per iteration).
struct Lanes { int lanesC, rowsPer; };
static Lanes lanes_for(int C)
{
    const int CP = (C + 1) / 2;
    int l = 1;
    while (l < CP && l < 256) l <<= 1;
    return Lanes{l, 256 / l};
}
static int reduce_blocks(int M, int rowsPer)
{
    long b = ((long)M + (long)rowsPer * 4 - 1) / ((long)rowsPer * 4);      // >= 4 rows per row-lane
    static const int gmax = getenv("YN_RED_G") ? atoi(getenv("YN_RED_G")) : 512;
    if (b > gmax) b = gmax;
    if (b < 1) b = 1;
    return (int)b;
}
static int stream_blocks(int M, int rowsPer)
{
    long b = ((long)M + (long)rowsPer * 4 - 1) / ((long)rowsPer * 4);
    if (b > 256 * 8) b = 256 * 8;
    if (b < 1) b = 1;
    return (int)b;
}

struct RedArgs {
    const float* y; int y_ld, y_off;                  // stats / col-sum: the matrix; bwd sums: the pre-BN conv output (dense, ld = C)
    const float* dz; int dz_ld, dz_off, dz_cs;        // bwd sums: upstream gradient (may be a strided channel view)
    const float* mean; const float* invstd; const float* gamma; const float* beta;   // bwd sums (act' from the recomputed BN value)
    double* acc; float* facc; size_t slot_stride;
    int M, C, act, lanesC;
};

template <bool VEC> __device__ __forceinline__ float2 ld2(const float* p, int cs)
{
    if (VEC) return *reinterpret_cast<const float2*>(p);
    float2 v;
    v.x = p[0]; v.y = p[cs];
    return v;
}
// value if ok else 0, as a bit mask: keeps the (always legal, clamped-address) load unconditional — with a ?: select
// the compiler sinks the load into a branch and waits for every load separately
__device__ __forceinline__ float keep_if(float v, bool ok)
{
    unsigned mk = ok ? 0xffffffffu : 0u;
    asm volatile("" : "+v"(mk));                            // opaque: otherwise `v & mask` is recognised as a select again
    return __uint_as_float(__float_as_uint(v) & mk);
}
__device__ __forceinline__ float act_grad(float g, float zz, int act) { return zz > 0.0f ? g : (act == 2 ? 0.1f * g : 0.0f); }
// BatchNorm output before the activation, with a FIXED rounding sequence: the backward kernels recompute it from the saved
// conv output to get the activation's sign instead of reading the activation output back from memory.
__device__ __forceinline__ float bn_value(float y, float mu, float is, float ga, float be)
{
    return __fmaf_rn(__fmul_rn(__fsub_rn(y, mu), is), ga, be);
}
__device__ __forceinline__ double acc_sum(const double* acc, int C, int which, int c)
{
    double v = 0.0;
#pragma unroll
    for (int s = 0; s < ACC_SLOTS; ++s) v += acc[((size_t)s * 2 + which) * C + c];
    return v;
}

// MODE 0: stats (sum, sum of squares) -> double atomics ; MODE 2: BN backward sums -> double atomics ; MODE 3: plain column sum -> float atomics
template <int MODE, bool YVEC, bool DVEC>
__global__ __launch_bounds__(256) void col_reduce_kernel(RedArgs a)
{
    __shared__ double red[256][4];
    constexpr int U = 4;
    const int lanesC = a.lanesC, rowsPer = 256 / lanesC;
    const int cl = threadIdx.x & (lanesC - 1), rl = threadIdx.x / lanesC;
    const int CP = (a.C + 1) >> 1;
    for (int cp = cl; cp < ((CP + lanesC - 1) / lanesC) * lanesC; cp += lanesC) {
        const int c0 = cp * 2;
        const bool live = cp < CP, has1 = c0 + 1 < a.C;
        double s0a = 0.0, s0b = 0.0, s1a = 0.0, s1b = 0.0;
        if (live) {
            float mu0 = 0.0f, mu1 = 0.0f, is0 = 0.0f, is1 = 0.0f, ga0 = 0.0f, ga1 = 0.0f, be0 = 0.0f, be1 = 0.0f;
            if (MODE == 2) {
                const int c1 = has1 ? c0 + 1 : c0;
                mu0 = a.mean[c0]; is0 = a.invstd[c0]; ga0 = a.gamma[c0]; be0 = a.beta[c0];
                mu1 = a.mean[c1]; is1 = a.invstd[c1]; ga1 = a.gamma[c1]; be1 = a.beta[c1];
            }
            const float* yb = a.y + a.y_off + c0;
            const float* db = MODE == 2 ? a.dz + a.dz_off + (size_t)c0 * a.dz_cs : nullptr;
            auto consume = [&](float2 v, float2 g) {
                if (!has1) { v.y = 0.0f; g.y = 0.0f; }
                if (MODE == 0) {
                    s0a += (double)v.x; s0b += (double)v.y;
                    s1a += (double)v.x * (double)v.x; s1b += (double)v.y * (double)v.y;
                } else if (MODE == 3) {
                    s0a += (double)v.x; s0b += (double)v.y;
                } else {
                    if (a.act) {
                        g.x = act_grad(g.x, bn_value(v.x, mu0, is0, ga0, be0), a.act);
                        g.y = act_grad(g.y, bn_value(v.y, mu1, is1, ga1, be1), a.act);
                    }
                    const float xh0 = (v.x - mu0) * is0, xh1 = (v.y - mu1) * is1;
                    s0a += (double)g.x; s0b += (double)g.y;
                    s1a += (double)g.x * (double)xh0; s1b += (double)g.y * (double)xh1;
                }
            };
            const long step = (long)gridDim.x * rowsPer;
            long r = (long)blockIdx.x * rowsPer + rl;
            for (; r + (U - 1) * step < a.M; r += U * step) {
                float2 v[U], g[U];
#pragma unroll
                for (int u = 0; u < U; ++u) v[u] = ld2<YVEC>(yb + (size_t)(r + u * step) * a.y_ld, 1);
#pragma unroll
                for (int u = 0; u < U; ++u) g[u] = MODE == 2 ? ld2<DVEC>(db + (size_t)(r + u * step) * a.dz_ld, a.dz_cs) : make_float2(0.0f, 0.0f);
#pragma unroll
                for (int u = 0; u < U; ++u) consume(v[u], g[u]);
            }
            for (; r < a.M; r += step) {
                const float2 v = ld2<YVEC>(yb + (size_t)r * a.y_ld, 1);
                const float2 g = MODE == 2 ? ld2<DVEC>(db + (size_t)r * a.dz_ld, a.dz_cs) : make_float2(0.0f, 0.0f);
                consume(v, g);
            }
        }
        __syncthreads();
        red[threadIdx.x][0] = s0a; red[threadIdx.x][1] = s0b; red[threadIdx.x][2] = s1a; red[threadIdx.x][3] = s1b;
        __syncthreads();
        if (rl == 0 && live) {
            for (int k = 1; k < rowsPer; ++k) {
                const double* q = red[k * lanesC + cl];
                s0a += q[0]; s0b += q[1]; s1a += q[2]; s1b += q[3];
            }
            // same-address atomics serialise at ~50 ns each: spread the blocks over several copies of the accumulator
            if (MODE == 3) {
                float* f = a.facc + (size_t)(blockIdx.x & (GRAD_SLOTS - 1)) * a.slot_stride;
                atomicAdd(f + c0, (float)s0a);
                if (has1) atomicAdd(f + c0 + 1, (float)s0b);
            } else {
                double* acc = a.acc + (size_t)(blockIdx.x & (ACC_SLOTS - 1)) * 2 * a.C;
                atomicAdd(acc + c0, s0a); atomicAdd(acc + a.C + c0, s1a);
                if (has1) { atomicAdd(acc + c0 + 1, s0b); atomicAdd(acc + a.C + c0 + 1, s1b); }
            }
        }
    }
}

template <int MODE>
static void launch_col_reduce(const RedArgs& a, hipStream_t s)
{
    const Lanes L = lanes_for(a.C);
    const dim3 grid(reduce_blocks(a.M, L.rowsPer));
    const bool yvec = !(a.y_ld & 1) && !(a.y_off & 1);
    const bool dvec = MODE == 2 && a.dz_cs == 1 && !(a.dz_ld & 1) && !(a.dz_off & 1);
    if (yvec && dvec) hipLaunchKernelGGL((col_reduce_kernel<MODE, true, true>), grid, dim3(256), 0, s, a);
    else if (yvec) hipLaunchKernelGGL((col_reduce_kernel<MODE, true, false>), grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL((col_reduce_kernel<MODE, false, false>), grid, dim3(256), 0, s, a);
}

void launch_bn_stats(const float* y, int M, int C, double* acc, hipStream_t s)
{
    RedArgs a{};
    a.y = y; a.y_ld = C; a.y_off = 0; a.acc = acc; a.M = M; a.C = C; a.lanesC = lanes_for(C).lanesC;
    launch_col_reduce<0>(a, s);
}

// out[c] += sum_m x[m*ld + off + c]   (float atomics into the gradient slots; every pair (c, c+1) must be readable: ld > C or C even)
void launch_col_sum_accumulate(const float* x, int ld, int off, int M, int C, float* out, size_t slot_stride, hipStream_t s)
{
    RedArgs a{};
    a.y = x; a.y_ld = ld; a.y_off = off; a.facc = out; a.slot_stride = slot_stride; a.M = M; a.C = C; a.lanesC = lanes_for(C).lanesC;
    launch_col_reduce<3>(a, s);
}

// ---- BatchNorm forward apply: z = act((y - mean) * invstd * gamma + beta) with mean / invstd derived from the stats
//      accumulator; optional channel-interleaved output (out[m][off + c*cs]) and pass-through copy
//      (out[m][pass_dst_off + c*cs] = pass[m][pass_off + c]) = concat + channel shuffle.  Block 0 also saves mean / invstd
//      for the backward pass and updates the running statistics (momentum 0.1, unbiased variance).
//      OUT: 0 generic scalar stores, 1 float2 stores (dense output), 2 float4 stores of the shuffled pair {pass, z, pass, z}.
template <int OUT>
__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyArgs a)
{
    constexpr int U = 4;
    const int lanesC = a.lanesC, rowsPer = 256 / lanesC;
    const int cl = threadIdx.x & (lanesC - 1), rl = threadIdx.x / lanesC;
    const int CP = (a.C + 1) >> 1;
    const double invM = 1.0 / (double)a.M;
    for (int cp = cl; cp < CP; cp += lanesC) {
        const int c0 = cp * 2;
        const bool has1 = c0 + 1 < a.C;
        float mu[2], is[2], ga[2], be[2];
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + j < a.C ? c0 + j : c0;
            const double m = acc_sum(a.acc, a.C, 0, c) * invM;
            double var = acc_sum(a.acc, a.C, 1, c) * invM - m * m;
            if (var < 0.0) var = 0.0;
            mu[j] = (float)m; is[j] = (float)(1.0 / sqrt(var + (double)a.eps)); ga[j] = a.gamma[c]; be[j] = a.beta[c];
            if (blockIdx.x == 0 && rl == 0 && (j == 0 || has1)) {
                a.mean[c] = mu[j]; a.invstd[c] = is[j];
                if (a.rmean) {
                    const float unbiased = (float)(a.M > 1 ? var * ((double)a.M / (double)(a.M - 1)) : var);
                    a.rmean[c] = (1.0f - a.momentum) * a.rmean[c] + a.momentum * mu[j];
                    a.rvar[c] = (1.0f - a.momentum) * a.rvar[c] + a.momentum * unbiased;
                }
            }
        }
        const float* yb = a.y + c0;
        const float* pb = a.pass ? a.pass + a.pass_off + c0 : nullptr;
        auto emit = [&](long r, float2 v, float2 p) {
            v.x = bn_value(v.x, mu[0], is[0], ga[0], be[0]);
            v.y = bn_value(v.y, mu[1], is[1], ga[1], be[1]);
            if (a.act == 1) { v.x = v.x > 0.0f ? v.x : 0.0f; v.y = v.y > 0.0f ? v.y : 0.0f; }
            else if (a.act == 2) { v.x = v.x > 0.0f ? v.x : 0.1f * v.x; v.y = v.y > 0.0f ? v.y : 0.1f * v.y; }
            float* o = a.out + (size_t)r * a.out_ld;
            if (OUT == 2) *reinterpret_cast<float4*>(o + 2 * c0) = make_float4(p.x, v.x, p.y, v.y);
            else if (OUT == 1) *reinterpret_cast<float2*>(o + a.out_off + c0) = v;
            else {
                o[a.out_off + (size_t)c0 * a.out_cs] = v.x;
                if (has1) o[a.out_off + (size_t)(c0 + 1) * a.out_cs] = v.y;
                if (pb) {
                    o[a.pass_dst_off + (size_t)c0 * a.out_cs] = p.x;
                    if (has1) o[a.pass_dst_off + (size_t)(c0 + 1) * a.out_cs] = p.y;
                }
            }
        };
        const long step = (long)gridDim.x * rowsPer;
        long r = (long)blockIdx.x * rowsPer + rl;
        for (; r + (U - 1) * step < a.M; r += U * step) {
            float2 v[U], p[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = ld2<OUT != 0>(yb + (size_t)(r + u * step) * a.C, 1);
#pragma unroll
            for (int u = 0; u < U; ++u) p[u] = pb ? ld2<OUT == 2>(pb + (size_t)(r + u * step) * a.pass_ld, 1) : make_float2(0.0f, 0.0f);
#pragma unroll
            for (int u = 0; u < U; ++u) emit(r + u * step, v[u], p[u]);
        }
        for (; r < a.M; r += step) {
            const float2 v = ld2<OUT != 0>(yb + (size_t)r * a.C, 1);
            const float2 p = pb ? ld2<OUT == 2>(pb + (size_t)r * a.pass_ld, 1) : make_float2(0.0f, 0.0f);
            emit(r, v, p);
        }
    }
}

void launch_bn_apply(const BnApplyArgs& a0, hipStream_t s)
{
    BnApplyArgs a = a0;
    const Lanes L = lanes_for(a.C);
    a.lanesC = L.lanesC;
    const dim3 grid(stream_blocks(a.M, L.rowsPer));
    const bool even = !(a.C & 1);
    const bool shuf = even && a.pass && a.out_cs == 2 && a.out_off == 1 && a.pass_dst_off == 0 && !(a.out_ld & 3) && !(a.pass_ld & 1) && !(a.pass_off & 1);
    const bool ovec = even && !a.pass && a.out_cs == 1 && !(a.out_ld & 1) && !(a.out_off & 1);
    if (shuf) hipLaunchKernelGGL(bn_apply_kernel<2>, grid, dim3(256), 0, s, a);
    else if (ovec) hipLaunchKernelGGL(bn_apply_kernel<1>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(bn_apply_kernel<0>, grid, dim3(256), 0, s, a);
}

// ---- BatchNorm backward: dy = gamma * invstd * (dyh - mean(dyh) - xhat * mean(dyh * xhat)),  dyh = dz * act'(z);
//      block 0 writes dbeta = sum dyh and dgamma = sum dyh * xhat.
template <bool DVEC>
__global__ __launch_bounds__(256) void bn_bwd_kernel(BnBwdArgs a)
{
    constexpr int U = 4;
    const int lanesC = a.lanesC, rowsPer = 256 / lanesC;
    const int cl = threadIdx.x & (lanesC - 1), rl = threadIdx.x / lanesC;
    const int CP = (a.C + 1) >> 1;
    const double invM = 1.0 / (double)a.M;
    const bool yvec = !(a.C & 1);
    for (int cp = cl; cp < CP; cp += lanesC) {
        const int c0 = cp * 2;
        const bool has1 = c0 + 1 < a.C;
        float mu[2], is[2], ga[2], be[2], k[2], m0[2], m1[2];
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + j < a.C ? c0 + j : c0;
            mu[j] = a.mean[c]; is[j] = a.invstd[c]; ga[j] = a.gamma[c]; be[j] = a.beta[c]; k[j] = ga[j] * is[j];
            const double s0 = acc_sum(a.acc, a.C, 0, c), s1 = acc_sum(a.acc, a.C, 1, c);
            m0[j] = (float)(s0 * invM); m1[j] = (float)(s1 * invM);
            if (blockIdx.x == 0 && rl == 0 && (j == 0 || has1)) { a.dbeta[c] = (float)s0; a.dgamma[c] = (float)s1; }
        }
        const float* yb = a.y + c0;
        const float* db = a.dz + a.dz_off + (size_t)c0 * a.dz_cs;
        auto emit = [&](long r, float2 v, float2 g) {
            if (a.act) {
                g.x = act_grad(g.x, bn_value(v.x, mu[0], is[0], ga[0], be[0]), a.act);
                g.y = act_grad(g.y, bn_value(v.y, mu[1], is[1], ga[1], be[1]), a.act);
            }
            const float xh0 = (v.x - mu[0]) * is[0], xh1 = (v.y - mu[1]) * is[1];
            g.x = k[0] * (g.x - m0[0] - xh0 * m1[0]);
            g.y = k[1] * (g.y - m0[1] - xh1 * m1[1]);
            float* o = a.dy + (size_t)r * a.C + c0;
            if (yvec) *reinterpret_cast<float2*>(o) = g;
            else { o[0] = g.x; if (has1) o[1] = g.y; }
        };
        const long step = (long)gridDim.x * rowsPer;
        long r = (long)blockIdx.x * rowsPer + rl;
        for (; r + (U - 1) * step < a.M; r += U * step) {
            float2 v[U], g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = yvec ? ld2<true>(yb + (size_t)(r + u * step) * a.C, 1) : ld2<false>(yb + (size_t)(r + u * step) * a.C, 1);
#pragma unroll
            for (int u = 0; u < U; ++u) g[u] = ld2<DVEC>(db + (size_t)(r + u * step) * a.dz_ld, a.dz_cs);
#pragma unroll
            for (int u = 0; u < U; ++u) emit(r + u * step, v[u], g[u]);
        }
        for (; r < a.M; r += step) {
            const float2 v = yvec ? ld2<true>(yb + (size_t)r * a.C, 1) : ld2<false>(yb + (size_t)r * a.C, 1);
            emit(r, v, ld2<DVEC>(db + (size_t)r * a.dz_ld, a.dz_cs));
        }
    }
}

void launch_bn_bwd(const BnBwdArgs& a0, hipStream_t s)
{
    BnBwdArgs a = a0;
    const Lanes L = lanes_for(a.C);
    a.lanesC = L.lanesC;
    RedArgs c{};
    c.y = a.y; c.y_ld = a.C; c.y_off = 0;
    c.dz = a.dz; c.dz_ld = a.dz_ld; c.dz_off = a.dz_off; c.dz_cs = a.dz_cs;
    c.mean = a.mean; c.invstd = a.invstd; c.gamma = a.gamma; c.beta = a.beta; c.acc = a.acc; c.M = a.M; c.C = a.C; c.act = a.act; c.lanesC = L.lanesC;
    launch_col_reduce<2>(c, s);
    const bool dvec = a.dz_cs == 1 && !(a.dz_ld & 1) && !(a.dz_off & 1);
    const dim3 grid(stream_blocks(a.M, L.rowsPer));
    if (dvec) hipLaunchKernelGGL(bn_bwd_kernel<true>, grid, dim3(256), 0, s, a);
    else hipLaunchKernelGGL(bn_bwd_kernel<false>, grid, dim3(256), 0, s, a);
}

// =================================================================================================
// Weight gradient of a GEMM-shaped conv:  dW[n][k] += sum_m dY[m][n] * X[m][k]      (n < N = Cout, k < K)
//   pointwise: X[m][k] = x[m*ld + off + k];   dense 3x3: k = tap*Cin + ci, X = im2col of x (zero outside the image).
// Output is written in the reference's weight layout: pointwise [Cout][Cin]; dense [Cout][Cin][3][3].
//
// No LDS staging: the 32x32x2 f32 MFMA takes one A and one B value per lane, and lane l of a wave reads element l%32 of
// row 2*step + l/32 of dY / X straight from global memory (two 128-byte row segments per load instruction).  A wave owns
// an (NT x KT) grid of 32x32 accumulators, so one pair of rows costs NT + KT loads for NT*KT MFMAs.  The four waves of a
// block interleave over the row pairs of the block's M slice (grid.z); they are combined through LDS tile by tile and
// added with float atomics into the block's gradient slot.
// =================================================================================================
template <int NT, int KT, bool DENSE>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a)
{
    __shared__ float red[3][64][17];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hh = lane >> 5;
    const int n0 = blockIdx.x * (NT * 32), k0 = blockIdx.y * (KT * 32);
    const int slices = gridDim.z;
    const int rows = (((a.M + slices - 1) / slices) + 7) & ~7;
    const int m_begin = blockIdx.z * rows, m_end = min(a.M, m_begin + rows);
    f32x16 acc[NT][KT];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
    // per-lane column bookkeeping
    const float* dyp[NT]; bool nok[NT];
#pragma unroll
    for (int i = 0; i < NT; ++i) { const int n = n0 + i * 32 + l31; nok[i] = n < a.N; dyp[i] = a.dy + (nok[i] ? n : 0); }
    const float* xp[KT]; bool kok[KT]; int tdy[KT], tdx[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) {
        const int k = k0 + j * 32 + l31;
        kok[j] = k < a.K;
        const int kk = kok[j] ? k : 0;
        if (DENSE) {
            const int tap = kk / a.Cin, ci = kk - tap * a.Cin;
            tdy[j] = tap / 3 - 1; tdx[j] = tap % 3 - 1;
            xp[j] = a.x + (ptrdiff_t)(tdy[j] * a.W + tdx[j]) * a.x_ld + a.x_off + ci;
        } else {
            tdy[j] = tdx[j] = 0;
            xp[j] = a.x + a.x_off + kk;
        }
    }
    const int hw = a.H * a.W;
    constexpr int U = 4;                                   // row pairs loaded before any MFMA is issued
    ptrdiff_t toff[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) toff[j] = DENSE ? (ptrdiff_t)(tdy[j] * a.W + tdx[j]) * a.x_ld : 0;
    if (DENSE) {
#pragma unroll
        for (int j = 0; j < KT; ++j) xp[j] -= toff[j];      // centre-tap pointer; the tap offset is added only when the tap is inside the image
    }
    // (a two-deep software pipeline over the batches was measured slower: most waves only own 3-12 batches)
    auto load_batch = [&](int m0, float (&av)[U][NT], float (&bv)[U][KT]) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int m = m0 + 8 * u + hh;
            const bool mok = m < m_end;
            const int mc = mok ? m : m_begin;               // clamped row: the load is always legal, the value is masked off
            int py = 0, px = 0;
            if (DENSE) { const int rem = mc % hw; py = rem / a.W; px = rem - py * a.W; }
#pragma unroll
            for (int i = 0; i < NT; ++i) av[u][i] = keep_if(dyp[i][(size_t)mc * a.dy_ld], mok && nok[i]);
#pragma unroll
            for (int j = 0; j < KT; ++j) {
                const bool ok = mok && kok[j];
                bool inside = true;
                if (DENSE) inside = (unsigned)(py + tdy[j]) < (unsigned)a.H && (unsigned)(px + tdx[j]) < (unsigned)a.W;
                bv[u][j] = keep_if(xp[j][(ptrdiff_t)((size_t)mc * a.x_ld) + (inside ? toff[j] : 0)], ok && inside);
            }
        }
    };
    auto mfma_batch = [&](const float (&av)[U][NT], const float (&bv)[U][KT]) {
#pragma unroll
        for (int u = 0; u < U; ++u)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < KT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u][i], bv[u][j], acc[i][j], 0, 0, 0);
    };
    for (int m0 = m_begin + 2 * wave; m0 < m_end; m0 += 8 * U) {
        float av[U][NT], bv[U][KT];
        load_batch(m0, av, bv);
        __builtin_amdgcn_sched_barrier(0);                  // all loads of the batch are in flight before the first MFMA waits
        mfma_batch(av, bv);
    }
    // combine the four waves tile by tile; acc[..][r]: row i = (r&3) + 8*(r>>2) + 4*hh (n index), column l31 (k index)
    float* out = a.partial + (size_t)blockIdx.z * ((size_t)a.N * a.K);
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < KT; ++j) {
            __syncthreads();
            if (wave > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave - 1][lane][r] = acc[i][j][r];
            }
            __syncthreads();
            if (wave == 0) {
                const int k = k0 + j * 32 + l31;
                if (k < a.K) {
                    int tap = 0, ci = k;
                    if (DENSE) { tap = k / a.Cin; ci = k - tap * a.Cin; }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int n = n0 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * hh;
                        if (n >= a.N) continue;
                        const float v = acc[i][j][r] + red[0][lane][r] + red[1][lane][r] + red[2][lane][r];
                        const size_t idx = DENSE ? ((size_t)n * a.Cin + ci) * 9 + tap : (size_t)n * a.K + k;
                        out[idx] = v;
                    }
                }
            }
        }
}

// dW[i] = sum over slices of partial[s][i], in slice order (deterministic)
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ partial, int slices, long n, float* __restrict__ dw)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s0 = 0.0f, s1 = 0.0f, s2 = 0.0f, s3 = 0.0f;
    int s = 0;
    for (; s + 3 < slices; s += 4) {
        s0 += partial[(size_t)s * n + i]; s1 += partial[(size_t)(s + 1) * n + i];
        s2 += partial[(size_t)(s + 2) * n + i]; s3 += partial[(size_t)(s + 3) * n + i];
    }
    for (; s < slices; ++s) s0 += partial[(size_t)s * n + i];
    dw[i] = (s0 + s1) + (s2 + s3);
}

template <int NT, int KT>
static void launch_wgrad_t(const WgradArgs& a, hipStream_t s)
{
    const int gn = (a.N + NT * 32 - 1) / (NT * 32), gk = (a.K + KT * 32 - 1) / (KT * 32);
    // M slices: enough blocks for one wave per SIMD (three for the 2x2 tile, whose accumulators leave room for three
    // resident waves); more slices only add per-slice copies of dW to write and re-read.  YN_WG_SLICES overrides (tuning).
    static const int senv = getenv("YN_WG_SLICES") ? atoi(getenv("YN_WG_SLICES")) : 0;
    const int smax = senv > 0 ? senv : (NT * KT <= 4 ? 768 : 256);
    int slices = smax / (gn * gk);
    const int max_slices = (a.M + 255) / 256;
    if (slices > max_slices) slices = max_slices;
    const long nk = (long)a.N * a.K;
    if ((long)slices * nk > (long)a.partial_cap) slices = (int)((long)a.partial_cap / nk);
    if (slices < 1) slices = 1;
    if (a.dense) hipLaunchKernelGGL((wgrad_kernel<NT, KT, true>), dim3(gn, gk, slices), dim3(256), 0, s, a);
    else hipLaunchKernelGGL((wgrad_kernel<NT, KT, false>), dim3(gn, gk, slices), dim3(256), 0, s, a);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((nk + 255) / 256)), dim3(256), 0, s, a.partial, slices, nk, a.dw);
}

void launch_wgrad(const WgradArgs& a, hipStream_t s)
{
    auto tiles = [&](int nt, int kt) { return (long)((a.N + nt * 32 - 1) / (nt * 32)) * nt * ((a.K + kt * 32 - 1) / (kt * 32)) * kt; };
    const long c22 = tiles(2, 2), c24 = tiles(2, 4), c33 = tiles(3, 3);
    if (c33 <= c24 && c33 <= c22) launch_wgrad_t<3, 3>(a, s);
    else if (c24 <= c22) launch_wgrad_t<2, 4>(a, s);
    else launch_wgrad_t<2, 2>(a, s);
}

// ---- depthwise 3x3 weight gradient: dW[c][tap] += sum_p dY[p][c] * X[p*stride + tap - 1][c]  (torch layout [C][1][3][3])
//      Same lane layout as the column reductions: a lane owns two channels (float2 loads); a row-lane takes RUNS of
//      output pixels along an image row (8 at stride 1, 4 at stride 2), loads the whole input window of the run
//      (3 rows x 10 / 9 columns) and the run's dY with unconditional clamped loads, then does the 18 x RUN FMAs from
//      registers.  LDS combine over the row-lanes, one float atomic per (c, tap) per block into the block's gradient slot.
template <int STRIDE, bool XVEC>
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int x_ld, int x_off,
                                                        int B, int H, int W, int C, float* __restrict__ part, int lanesC)
{
    __shared__ float red[256][19];
    constexpr int RUN = STRIDE == 1 ? 8 : 4;
    constexpr int NCOL = STRIDE == 1 ? RUN + 2 : 2 * RUN + 1;
    const int Ho = (H - 1) / STRIDE + 1, Wo = (W - 1) / STRIDE + 1;
    const int runsPerRow = (Wo + RUN - 1) / RUN, totalRuns = B * Ho * runsPerRow;
    const int rowsPer = 256 / lanesC;
    const int cl = threadIdx.x & (lanesC - 1), rl = threadIdx.x / lanesC;
    const int CP = (C + 1) >> 1;
    for (int cp = cl; cp < ((CP + lanesC - 1) / lanesC) * lanesC; cp += lanesC) {
        const int c0 = cp * 2;
        const bool live = cp < CP, has1 = c0 + 1 < C;
        float acc[18];
#pragma unroll
        for (int k = 0; k < 18; ++k) acc[k] = 0.0f;
        if (live) {
            // a block owns a CONTIGUOUS range of runs, and blocks are ordered XCD-contiguously: vertically adjacent runs share two
            // of their three input rows, which a strided assignment fetched into every XCD's L2 (3x the tensor)
            const int per = (totalRuns + (int)gridDim.x * rowsPer - 1) / ((int)gridDim.x * rowsPer);
            const int ubase = (int)xcd_block(blockIdx.x, gridDim.x) * per * rowsPer;
            for (int it = 0; it < per; ++it) {
                const int u = ubase + it * rowsPer + rl;
                if (u >= totalRuns) break;
                const int row = u / runsPerRow, seg = u - row * runsPerRow;
                const int b = row / Ho, oy = row - b * Ho;
                const int ox0 = seg * RUN;
                float2 col[NCOL][3], g[RUN];
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int iy = oy * STRIDE - 1 + ky;
                    const bool yok = iy >= 0 && iy < H;
                    const float* xr = x + ((size_t)(b * H + (yok ? iy : 0)) * W) * x_ld + x_off + c0;
#pragma unroll
                    for (int j = 0; j < NCOL; ++j) {
                        const int ix = ox0 * STRIDE - 1 + j;
                        const bool ok = yok && ix >= 0 && ix < W;
                        const float2 v = ld2<XVEC>(xr + (size_t)(ix < 0 ? 0 : (ix >= W ? W - 1 : ix)) * x_ld, 1);
                        col[j][ky] = ok ? v : make_float2(0.0f, 0.0f);
                    }
                }
                const float* dyp = dy + ((size_t)row * Wo) * C + c0;
#pragma unroll
                for (int o = 0; o < RUN; ++o) {
                    const int ox = ox0 + o;
                    const float2 v = ld2<XVEC>(dyp + (size_t)(ox < Wo ? ox : Wo - 1) * C, 1);      // XVEC implies C even
                    g[o] = ox < Wo ? v : make_float2(0.0f, 0.0f);
                }
#pragma unroll
                for (int o = 0; o < RUN; ++o)
#pragma unroll
                    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                        for (int kx = 0; kx < 3; ++kx) {
                            acc[(ky * 3 + kx) * 2 + 0] += g[o].x * col[o * STRIDE + kx][ky].x;
                            acc[(ky * 3 + kx) * 2 + 1] += g[o].y * col[o * STRIDE + kx][ky].y;
                        }
            }
        }
        if (!has1) {
#pragma unroll
            for (int k = 0; k < 9; ++k) acc[2 * k + 1] = 0.0f;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 18; ++k) red[threadIdx.x][k] = acc[k];
        __syncthreads();
        if (rl == 0 && live) {
            for (int j = 1; j < rowsPer; ++j)
#pragma unroll
                for (int k = 0; k < 18; ++k) acc[k] += red[j * lanesC + cl][k];
            float* out = part + (size_t)blockIdx.x * C * 9;         // the block's own row of the scratch matrix (rows_sum_kernel adds the rows up)
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                out[(size_t)c0 * 9 + k] = acc[2 * k];
                if (has1) out[(size_t)(c0 + 1) * 9 + k] = acc[2 * k + 1];
            }
        }
    }
}

// out[i] += sum over the G rows of part[g][i]; grid (ceil(n / 256), GY): block row y sums its slice of g, one atomic per element
__global__ __launch_bounds__(256) void rows_sum_kernel(const float* __restrict__ part, int G, int n, float* __restrict__ out)
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
    if (g1 > g0) atomicAdd(out + i, (s0 + s1) + (s2 + s3));
}

void launch_dw_wgrad(const float* dy, const float* x, int x_ld, int x_off, int B, int H, int W, int C, int stride, float* dw, float* part, size_t part_cap, hipStream_t s)
{
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const int run = stride == 1 ? 8 : 4;
    const int runs = B * Ho * ((Wo + run - 1) / run);
    const Lanes L = lanes_for(C);
    // per-block partial rows instead of atomics into 8 slots (round 3, as the fp16 step's hdw_wgrad_kernel): the same-address chains had capped
    // the grid at 128 blocks
    static const int rpl = getenv("YN_DWW_RUNS") ? atoi(getenv("YN_DWW_RUNS")) : 2;
    int G = (runs + L.rowsPer * rpl - 1) / (L.rowsPer * rpl);
    static const int gmax = getenv("YN_DWW_G") ? atoi(getenv("YN_DWW_G")) : 1024;
    if (G > gmax) G = gmax;
    if ((size_t)G * C * 9 > part_cap) G = (int)(part_cap / ((size_t)C * 9));
    if (G < 8) G = 8;
    G = (int)xcd_grid((unsigned)G);                         // multiple of 8 for xcd_block
    if ((size_t)G * C * 9 > part_cap) G -= 8;
    const bool vec = !(C & 1) && !(x_ld & 1) && !(x_off & 1);
#define YN_DWW(ST, V) hipLaunchKernelGGL((dw_wgrad_kernel<ST, V>), dim3(G), dim3(256), 0, s, dy, x, x_ld, x_off, B, H, W, C, part, L.lanesC)
    if (stride == 1) { if (vec) YN_DWW(1, true); else YN_DWW(1, false); }
    else { if (vec) YN_DWW(2, true); else YN_DWW(2, false); }
#undef YN_DWW
    const int n = C * 9;
    hipLaunchKernelGGL(rows_sum_kernel, dim3((unsigned)((n + 255) / 256), (unsigned)(G >= 64 ? 16 : 1)), dim3(256), 0, s, part, G, n, dw);
}

// ---- gradient slots -> the flat gradient buffer: g[i] += sum_s slots[s][i]
__global__ __launch_bounds__(256) void grad_combine_kernel(float* __restrict__ g, const float* __restrict__ slots, long n, size_t stride)
{
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        float v = g[i];
#pragma unroll
        for (int s = 0; s < GRAD_SLOTS; ++s) v += slots[(size_t)s * stride + i];
        g[i] = v;
    }
}

void launch_grad_combine(float* g, const float* slots, long n, size_t stride, hipStream_t s)
{
    long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(grad_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, s, g, slots, n, stride);
}

// ---- depthwise 3x3 stride-2 input gradient: dX[iy][ix][c] = sum_{ky,kx} dY[(iy+1-ky)/2][(ix+1-kx)/2][c] * w[ky][kx][c]
//      over the taps for which the division is exact and the output pixel exists.  w packed [9][C].  accumulate: dX += .
//      Along each axis an input coordinate i is touched by output o0 = (i+1)>>1 through tap k0 = i+1-2*o0 (0 or 1) and, when
//      k0 == 0, also by o0-1 through tap 2: at most 2 x 2 contributions, fetched with unconditional clamped loads.
//      thread = (input pixel, channel pair).
__global__ __launch_bounds__(256) void dw_dgrad_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w, int B, int H, int W, int C,
                                                           float* __restrict__ dx, int dx_ld, int dx_off, int accumulate)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int cpn = C >> 1;
    const int total = B * H * W * cpn;
    const int i = (int)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;     // XCD-contiguous (yn_internal.h): stencil neighbours share an L2
    if (i >= total) return;
    const int cp = i % cpn;
    int p = i / cpn;
    const int ix = p % W; int q = p / W;
    const int iy = q % H, b = q / H;
    const int c = cp * 2;
    int oy[2], ky[2], ox[2], kx[2];
    bool vy[2], vx[2];
    oy[0] = (iy + 1) >> 1; ky[0] = iy + 1 - 2 * oy[0]; vy[0] = oy[0] < Ho;
    oy[1] = oy[0] - 1;     ky[1] = 2;                  vy[1] = ky[0] == 0 && oy[1] >= 0;
    ox[0] = (ix + 1) >> 1; kx[0] = ix + 1 - 2 * ox[0]; vx[0] = ox[0] < Wo;
    ox[1] = ox[0] - 1;     kx[1] = 2;                  vx[1] = kx[0] == 0 && ox[1] >= 0;
    float2 g[4], wv[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int a = u >> 1, e = u & 1;
        const bool ok = vy[a] && vx[e];
        const int oyc = vy[a] ? oy[a] : 0, oxc = vx[e] ? ox[e] : 0;
        const float2 v = *reinterpret_cast<const float2*>(dy + ((size_t)(b * Ho + oyc) * Wo + oxc) * C + c);
        unsigned mk = ok ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(mk));
        g[u] = make_float2(__uint_as_float(__float_as_uint(v.x) & mk), __uint_as_float(__float_as_uint(v.y) & mk));
        wv[u] = *reinterpret_cast<const float2*>(w + (ky[a] * 3 + kx[e]) * C + c);
    }
    float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int u = 0; u < 4; ++u) { acc.x += g[u].x * wv[u].x; acc.y += g[u].y * wv[u].y; }
    float2* d = reinterpret_cast<float2*>(dx + (size_t)p * dx_ld + dx_off + c);
    if (accumulate) { const float2 o = *d; acc.x += o.x; acc.y += o.y; }
    *d = acc;
}

void launch_dw_dgrad_s2(const float* dy, const float* w, int B, int H, int W, int C, float* dx, int dx_ld, int dx_off, int accumulate, hipStream_t s)
{
    const long total = (long)B * H * W * (C >> 1);
    hipLaunchKernelGGL(dw_dgrad_s2_kernel, dim3(xcd_grid((unsigned)((total + 255) / 256))), dim3(256), 0, s, dy, w, B, H, W, C, dx, dx_ld, dx_off, accumulate);
}

// ---- stem weight gradient: dW[co][ci][ky][kx] += sum_p dY[p][co] * x_nchw[b][ci][2oy-1+ky][2ox-1+kx]
//      As a GEMM: dW^T[r = ci*9+ky*3+kx (27 -> 32)][co (24 -> 32)] = Xpatch^T[32 x M] * dY[M x 32], one 32x32 f32 MFMA tile per
//      wave fed straight from global memory (lane l supplies patch element r = l%32 and dY column co = l%32 of output pixel
//      2*step + l/32).  A wave owns whole output rows; the four waves of a block are combined through LDS before the
//      float atomics into dW.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int B, int H, int W, int Cout,
                                                          float* __restrict__ partial)
{
    __shared__ float red[3][64][17];
    constexpr int U = 8;                                   // MFMA steps whose operands are loaded before any is issued
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, hh = lane >> 5;
    const int r = l31, ci = r / 9, ky = (r % 9) / 3, kx = r % 3;
    const bool rlive = r < 27, clive = l31 < Cout;
    const int nrows = B * Ho;
    f32x16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.0f;
    for (int row = blockIdx.x * 4 + wave; row < nrows; row += gridDim.x * 4) {
        const int b = row / Ho, oy = row - b * Ho;
        const int iy = 2 * oy - 1 + ky;
        const bool yok = rlive && iy >= 0 && iy < H;
        const float* xrow = x + (((size_t)b * 3 + (rlive ? ci : 0)) * H + (yok ? iy : 0)) * W;
        const float* drow = dy + (size_t)row * Wo * Cout + (clive ? l31 : 0);
        for (int ox0 = 0; ox0 < Wo; ox0 += 2 * U) {
            float av[U], bv[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ox = ox0 + 2 * u + hh;
                const int ix = 2 * ox + kx - 1;
                const bool ook = ox < Wo;
                const float xa = xrow[ix < 0 ? 0 : (ix >= W ? W - 1 : ix)];
                const float db = drow[(size_t)(ook ? ox : Wo - 1) * Cout];
                av[u] = keep_if(xa, yok && ook && ix >= 0 && ix < W);
                bv[u] = keep_if(db, clive && ook);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int u = 0; u < U; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
    }
    // acc[i]: row (patch element) = (i&3) + 8*(i>>2) + 4*hh, column (co) = l31
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave - 1][lane][i] = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
        float* out = partial + (size_t)blockIdx.x * ((size_t)Cout * 27);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = acc[i] + red[0][lane][i] + red[1][lane][i] + red[2][lane][i];
            const int rr = (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (rr < 27 && clive) out[(size_t)l31 * 27 + rr] = v;
        }
    }
}

void launch_stem_wgrad(const float* dy, const float* x, int B, int H, int W, int Cout, float* dw, float* partial, size_t partial_cap, hipStream_t s)
{
    const int nrows = B * ((H - 1) / 2 + 1);
    int G = (nrows + 3) / 4;
    static const int gmax = getenv("YN_STEM_G") ? atoi(getenv("YN_STEM_G")) : 2048;
    if (G > gmax) G = gmax;
    const long n = (long)Cout * 27;
    if ((long)G * n > (long)partial_cap) G = (int)((long)partial_cap / n);
    if (G < 1) G = 1;
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(G), dim3(256), 0, s, dy, x, B, H, W, Cout, partial);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, partial, G, n, dw);
}

// ---- 3x3 stride-2 max pool forward that also records the arg-max (first maximum in window scan order, as ATen) and its
//      backward (scatter by recorded index).
__global__ __launch_bounds__(256) void maxpool_idx_kernel(const float* __restrict__ x, int B, int H, int W, int C, float* __restrict__ y, int32_t* __restrict__ idx)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * Wo * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int ox = (int)(p % Wo);
        const long q = p / Wo;
        const int oy = (int)(q % Ho), b = (int)(q / Ho);
        float m = -INFINITY;
        int best = -1;
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
            if (iy < 0 || iy >= H) continue;
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                if (ix < 0 || ix >= W) continue;
                const float v = x[((size_t)(b * H + iy) * W + ix) * C + c];
                if (v > m || best < 0) { m = v; best = iy * W + ix; }
            }
        }
        y[i] = m;
        idx[i] = best;
    }
}

// gather form (no atomics, no pre-zeroed output): an input pixel lies in at most 2 x 2 pooling windows — the same index algebra
// as dw_dgrad_s2 — and receives the gradient of every window whose recorded arg-max is this pixel.  thread = (pixel, channel pair)
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ idx, int B, int H, int W, int C, float* __restrict__ dx)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int cpn = C >> 1;
    const int total = B * H * W * cpn;
    const int i = (int)xcd_block(blockIdx.x, gridDim.x) * 256 + threadIdx.x;     // XCD-contiguous (yn_internal.h): stencil neighbours share an L2
    if (i >= total) return;
    const int cp = i % cpn;
    int p = i / cpn;
    const int ix = p % W; int q = p / W;
    const int iy = q % H, b = q / H;
    const int c = cp * 2, me = iy * W + ix;
    int oy[2], ox[2];
    bool vy[2], vx[2];
    oy[0] = (iy + 1) >> 1; vy[0] = oy[0] < Ho; oy[1] = oy[0] - 1; vy[1] = (iy & 1) && oy[1] >= 0;
    ox[0] = (ix + 1) >> 1; vx[0] = ox[0] < Wo; ox[1] = ox[0] - 1; vx[1] = (ix & 1) && ox[1] >= 0;
    float2 g[4]; int2 id[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int a = u >> 1, e = u & 1;
        const size_t o = ((size_t)(b * Ho + (vy[a] ? oy[a] : 0)) * Wo + (vx[e] ? ox[e] : 0)) * C + c;
        g[u] = *reinterpret_cast<const float2*>(dy + o);
        id[u] = *reinterpret_cast<const int2*>(idx + o);
    }
    float2 acc = make_float2(0.0f, 0.0f);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const bool ok = vy[u >> 1] && vx[u & 1];
        acc.x += (ok && id[u].x == me) ? g[u].x : 0.0f;
        acc.y += (ok && id[u].y == me) ? g[u].y : 0.0f;
    }
    *reinterpret_cast<float2*>(dx + (size_t)p * C + c) = acc;
}

void launch_maxpool_idx(const float* x, int B, int H, int W, int C, float* y, int32_t* idx, hipStream_t s)
{
    long blocks = ((long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * C + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(maxpool_idx_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, B, H, W, C, y, idx);
}

void launch_maxpool_bwd(const float* dy, const int32_t* idx, int B, int H, int W, int C, float* dx, hipStream_t s)
{
    long blocks = ((long)B * H * W * (C >> 1) + 255) / 256;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(xcd_grid((unsigned)blocks)), dim3(256), 0, s, dy, idx, B, H, W, C, dx);
}

// ---- element-wise glue --------------------------------------------------------------------------------
// mode 0: out = a + up2(b)     (b is [B,H/2,W/2,C])      models/yolo_nano.py:291-292
// mode 1: out = a + down(b)    (b is [B,2H,2W,C])        models/yolo_nano.py:295-296
// mode 2: b_grad[B,H/2,W/2,C] += sum of the 4 children of g[B,H,W,C]        (backward of up2)
// mode 3: b_grad[B,2H,2W,C] at even pixels += g[B,H,W,C]                    (backward of down)
__global__ __launch_bounds__(256) void resample_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                                        int B, int H, int W, int C, int mode)
{
    if (mode <= 1) {
        const long total = (long)B * H * W * C;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int c = (int)(i % C);
            const long p = i / C;
            const int x = (int)(p % W);
            const long q = p / W;
            const int y = (int)(q % H), bb = (int)(q / H);
            size_t j;
            if (mode == 0) j = (((size_t)bb * (H >> 1) + (y >> 1)) * (W >> 1) + (x >> 1)) * C + c;
            else           j = (((size_t)bb * (H << 1) + (y << 1)) * (W << 1) + (x << 1)) * C + c;
            out[i] = a[i] + b[j];
        }
    } else if (mode == 2) {                  // out = grad of the low-res tensor [B,H/2,W/2,C]; a = g [B,H,W,C]
        const int h2 = H >> 1, w2 = W >> 1;
        const long total = (long)B * h2 * w2 * C;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int c = (int)(i % C);
            const long p = i / C;
            const int x = (int)(p % w2);
            const long q = p / w2;
            const int y = (int)(q % h2), bb = (int)(q / h2);
            const size_t base = (((size_t)bb * H + 2 * y) * W + 2 * x) * C + c;
            out[i] += (a[base] + a[base + C]) + (a[base + (size_t)W * C] + a[base + (size_t)W * C + C]);
        }
    } else {                                 // out = grad of the high-res tensor [B,2H,2W,C]; a = g [B,H,W,C]
        const long total = (long)B * H * W * C;
        for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
            const int c = (int)(i % C);
            const long p = i / C;
            const int x = (int)(p % W);
            const long q = p / W;
            const int y = (int)(q % H), bb = (int)(q / H);
            out[(((size_t)bb * (H << 1) + (y << 1)) * (W << 1) + (x << 1)) * C + c] += a[i];
        }
    }
}

void launch_resample(const float* a, const float* b, float* out, int B, int H, int W, int C, int mode, hipStream_t s)
{
    long n = (long)B * H * W * C;
    if (mode == 2) n /= 4;
    long blocks = (n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(resample_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a, b, out, B, H, W, C, mode);
}

// dst[m][dst_off + j*dst_cs] (+)= src[m][src_off + j*src_cs]   for j < n, m < M
__global__ __launch_bounds__(256) void strided_copy_kernel(const float* __restrict__ src, int src_ld, int src_off, int src_cs,
                                                            float* __restrict__ dst, int dst_ld, int dst_off, int dst_cs, long M, int n, int accumulate)
{
    const long total = M * n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int j = (int)(i % n);
        const long m = i / n;
        const float v = src[(size_t)m * src_ld + src_off + j * src_cs];
        float* d = dst + (size_t)m * dst_ld + dst_off + j * dst_cs;
        *d = accumulate ? *d + v : v;
    }
}

void launch_strided_copy(const float* src, int src_ld, int src_off, int src_cs, float* dst, int dst_ld, int dst_off, int dst_cs,
                         long M, int n, int accumulate, hipStream_t s)
{
    long blocks = (M * n + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(strided_copy_kernel, dim3((unsigned)blocks), dim3(256), 0, s, src, src_ld, src_off, src_cs, dst, dst_ld, dst_off, dst_cs, M, n, accumulate);
}

// ---- weight packing for the backward (input-gradient) convolutions, from the raw torch-layout weights:
//   kind 0 (pointwise)  Wp[(n/2)][k][n&1] = W[n][k]                        : dX = dY * W  is a pointwise conv N -> K
//   kind 2 (dense 3x3)  k' = tap'*Cout + co, tap' = 8 - tap ;  Wp[(k'/2)][ci][k'&1] = W[co][ci][tap]   (flip + transpose)
//   kind 1 (depthwise)  Wd[8 - tap][c] = W[c][tap]                          (flipped taps)
__global__ void pack_bwd_kernel(const float* __restrict__ w, int Cout, int Cin, int kind, int Npad, float* __restrict__ out)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (kind == 1) {
        if (i >= Cout * 9) return;
        const int c = i / 9, tap = i - c * 9;
        out[(8 - tap) * Cout + c] = w[i];
    } else if (kind == 0) {
        if (i >= Cout * Cin) return;
        const int n = i / Cin, k = i - n * Cin;                 // W[n][k]; backward GEMM: K' = Cout (index n), N' = Cin (index k)
        out[((size_t)(n >> 1) * Npad + k) * 2 + (n & 1)] = w[i];
    } else {
        if (i >= Cout * Cin * 9) return;
        const int co = i / (Cin * 9), r = i - co * Cin * 9;
        const int ci = r / 9, tap = r - ci * 9;
        const int kk = (8 - tap) * Cout + co;
        out[((size_t)(kk >> 1) * Npad + ci) * 2 + (kk & 1)] = w[i];
    }
}

void launch_pack_bwd(const float* w, int Cout, int Cin, int kind, int Npad, float* out, hipStream_t s)
{
    const int total = kind == 1 ? Cout * 9 : (kind == 0 ? Cout * Cin : Cout * Cin * 9);
    hipLaunchKernelGGL(pack_bwd_kernel, dim3((total + 255) / 256), dim3(256), 0, s, w, Cout, Cin, kind, Npad, out);
}

}  // namespace ynk

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
    long b = ((long)M + (long)rowsPer * 8 - 1) / ((long)rowsPer * 8);      // >= 8 rows per row-lane
    if (b > 512) b = 512;
    if (b < 1) b = 1;
    return (int)b;
}

struct RedArgs {
    const float* y; int y_ld, y_off;                  // stats / col-sum: the matrix; bwd sums: the pre-BN conv output (dense, ld = C)
    const float* dz; int dz_ld, dz_off, dz_cs;        // bwd sums: upstream gradient (may be a strided channel view)
    const float* z; int z_ld, z_off, z_cs;            // bwd sums: BN+act output (for act')
    const float* mean; const float* invstd;
    double* acc; float* facc;
    int M, C, act, lanesC;
};

__device__ __forceinline__ float2 load2(const float* base, size_t row_off, int c0, int cs, bool vec, bool has1)
{
    if (vec) return *reinterpret_cast<const float2*>(base + row_off + c0);
    float2 v;
    v.x = base[row_off + (size_t)c0 * cs];
    v.y = has1 ? base[row_off + (size_t)(c0 + 1) * cs] : 0.0f;
    return v;
}
__device__ __forceinline__ float act_grad(float g, float zz, int act) { return zz > 0.0f ? g : (act == 2 ? 0.1f * g : 0.0f); }

// MODE 0: stats (sum, sum of squares) -> double atomics ; MODE 2: BN backward sums -> double atomics ; MODE 3: plain column sum -> float atomics
template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(RedArgs a)
{
    __shared__ double red[256][4];
    const int lanesC = a.lanesC, rowsPer = 256 / lanesC;
    const int cl = threadIdx.x & (lanesC - 1), rl = threadIdx.x / lanesC;
    const int CP = (a.C + 1) >> 1;
    const bool yvec = !(a.y_ld & 1) && !(a.y_off & 1);
    const bool dvec = MODE == 2 && a.dz_cs == 1 && !(a.dz_ld & 1) && !(a.dz_off & 1);
    const bool zvec = MODE == 2 && a.z_cs == 1 && !(a.z_ld & 1) && !(a.z_off & 1);
    for (int cp = cl; cp < ((CP + lanesC - 1) / lanesC) * lanesC; cp += lanesC) {
        const int c0 = cp * 2;
        const bool live = cp < CP, has1 = c0 + 1 < a.C;
        double s0a = 0.0, s0b = 0.0, s1a = 0.0, s1b = 0.0;
        if (live) {
            float mu0 = 0.0f, mu1 = 0.0f, is0 = 0.0f, is1 = 0.0f;
            if (MODE == 2) { mu0 = a.mean[c0]; is0 = a.invstd[c0]; if (has1) { mu1 = a.mean[c0 + 1]; is1 = a.invstd[c0 + 1]; } }
            for (long r = (long)blockIdx.x * rowsPer + rl; r < a.M; r += (long)gridDim.x * rowsPer) {
                const float2 v = load2(a.y, (size_t)r * a.y_ld + a.y_off, c0, 1, yvec && has1, has1);
                if (MODE == 0) {
                    s0a += (double)v.x; s0b += (double)v.y;
                    s1a += (double)v.x * (double)v.x; s1b += (double)v.y * (double)v.y;
                } else if (MODE == 3) {
                    s0a += (double)v.x; s0b += (double)v.y;
                } else {
                    float2 g = load2(a.dz, (size_t)r * a.dz_ld + a.dz_off, c0, a.dz_cs, dvec && has1, has1);
                    if (a.act) {
                        const float2 zz = load2(a.z, (size_t)r * a.z_ld + a.z_off, c0, a.z_cs, zvec && has1, has1);
                        g.x = act_grad(g.x, zz.x, a.act); g.y = act_grad(g.y, zz.y, a.act);
                    }
                    const float xh0 = (v.x - mu0) * is0, xh1 = (v.y - mu1) * is1;
                    s0a += (double)g.x; s0b += (double)g.y;
                    s1a += (double)g.x * (double)xh0; s1b += (double)g.y * (double)xh1;
                }
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
            if (MODE == 3) {
                atomicAdd(a.facc + c0, (float)s0a);
                if (has1) atomicAdd(a.facc + c0 + 1, (float)s0b);
            } else {
                atomicAdd(a.acc + c0, s0a); atomicAdd(a.acc + a.C + c0, s1a);
                if (has1) { atomicAdd(a.acc + c0 + 1, s0b); atomicAdd(a.acc + a.C + c0 + 1, s1b); }
            }
        }
    }
}

void launch_bn_stats(const float* y, int M, int C, double* acc, hipStream_t s)
{
    RedArgs a{};
    const Lanes L = lanes_for(C);
    a.y = y; a.y_ld = C; a.y_off = 0; a.acc = acc; a.M = M; a.C = C; a.lanesC = L.lanesC;
    hipLaunchKernelGGL(col_reduce_kernel<0>, dim3(reduce_blocks(M, L.rowsPer)), dim3(256), 0, s, a);
}

// out[c] += sum_m x[m*ld + off + c]   (float atomics; `out` is a zero-initialised gradient slice)
void launch_col_sum_accumulate(const float* x, int ld, int off, int M, int C, float* out, hipStream_t s)
{
    RedArgs a{};
    const Lanes L = lanes_for(C);
    a.y = x; a.y_ld = ld; a.y_off = off; a.facc = out; a.M = M; a.C = C; a.lanesC = L.lanesC;
    hipLaunchKernelGGL(col_reduce_kernel<3>, dim3(reduce_blocks(M, L.rowsPer)), dim3(256), 0, s, a);
}

// ---- BatchNorm forward apply: z = act((y - mean) * invstd * gamma + beta) with mean / invstd derived from the stats
//      accumulator; optional channel-interleaved output (out[m][off + c*cs]) and pass-through copy
//      (out[m][pass_dst_off + c*cs] = pass[m][pass_off + c]) = concat + channel shuffle.  Block 0 also saves mean / invstd
//      for the backward pass and updates the running statistics (momentum 0.1, unbiased variance).
__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyArgs a)
{
    const int lanesC = a.lanesC, rowsPer = 256 / lanesC;
    const int cl = threadIdx.x & (lanesC - 1), rl = threadIdx.x / lanesC;
    const int CP = (a.C + 1) >> 1;
    const double invM = 1.0 / (double)a.M;
    const bool ovec = a.out_cs == 1 && !(a.out_ld & 1) && !(a.out_off & 1);
    const bool shuf = a.pass && a.out_cs == 2 && a.out_off == 1 && a.pass_dst_off == 0 && !(a.out_ld & 3) && !(a.pass_ld & 1) && !(a.pass_off & 1);
    for (int cp = cl; cp < CP; cp += lanesC) {
        const int c0 = cp * 2;
        const bool has1 = c0 + 1 < a.C;
        float mu[2], is[2], ga[2], be[2];
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + j < a.C ? c0 + j : c0;
            const double m = a.acc[c] * invM;
            double var = a.acc[a.C + c] * invM - m * m;
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
        for (long r = (long)blockIdx.x * rowsPer + rl; r < a.M; r += (long)gridDim.x * rowsPer) {
            float2 v = load2(a.y, (size_t)r * a.C, c0, 1, has1, has1);
            v.x = (v.x - mu[0]) * is[0] * ga[0] + be[0];
            v.y = (v.y - mu[1]) * is[1] * ga[1] + be[1];
            if (a.act == 1) { v.x = v.x > 0.0f ? v.x : 0.0f; v.y = v.y > 0.0f ? v.y : 0.0f; }
            else if (a.act == 2) { v.x = v.x > 0.0f ? v.x : 0.1f * v.x; v.y = v.y > 0.0f ? v.y : 0.1f * v.y; }
            float* o = a.out + (size_t)r * a.out_ld;
            if (shuf && has1) {
                const float2 p = *reinterpret_cast<const float2*>(a.pass + (size_t)r * a.pass_ld + a.pass_off + c0);
                *reinterpret_cast<float4*>(o + 2 * c0) = make_float4(p.x, v.x, p.y, v.y);
            } else {
                if (ovec && has1) *reinterpret_cast<float2*>(o + a.out_off + c0) = v;
                else { o[a.out_off + (size_t)c0 * a.out_cs] = v.x; if (has1) o[a.out_off + (size_t)(c0 + 1) * a.out_cs] = v.y; }
                if (a.pass) {
                    const float* p = a.pass + (size_t)r * a.pass_ld + a.pass_off;
                    o[a.pass_dst_off + (size_t)c0 * a.out_cs] = p[c0];
                    if (has1) o[a.pass_dst_off + (size_t)(c0 + 1) * a.out_cs] = p[c0 + 1];
                }
            }
        }
    }
}

static int stream_blocks(int M, int rowsPer)
{
    long b = ((long)M + (long)rowsPer * 4 - 1) / ((long)rowsPer * 4);
    if (b > 256 * 8) b = 256 * 8;
    if (b < 1) b = 1;
    return (int)b;
}

void launch_bn_apply(const BnApplyArgs& a0, hipStream_t s)
{
    BnApplyArgs a = a0;
    const Lanes L = lanes_for(a.C);
    a.lanesC = L.lanesC;
    hipLaunchKernelGGL(bn_apply_kernel, dim3(stream_blocks(a.M, L.rowsPer)), dim3(256), 0, s, a);
}

// ---- BatchNorm backward: dy = gamma * invstd * (dyh - mean(dyh) - xhat * mean(dyh * xhat)),  dyh = dz * act'(z);
//      block 0 writes dbeta = sum dyh and dgamma = sum dyh * xhat.
__global__ __launch_bounds__(256) void bn_bwd_kernel(BnBwdArgs a)
{
    const int lanesC = a.lanesC, rowsPer = 256 / lanesC;
    const int cl = threadIdx.x & (lanesC - 1), rl = threadIdx.x / lanesC;
    const int CP = (a.C + 1) >> 1;
    const double invM = 1.0 / (double)a.M;
    const bool dvec = a.dz_cs == 1 && !(a.dz_ld & 1) && !(a.dz_off & 1);
    const bool zvec = a.z_cs == 1 && !(a.z_ld & 1) && !(a.z_off & 1);
    for (int cp = cl; cp < CP; cp += lanesC) {
        const int c0 = cp * 2;
        const bool has1 = c0 + 1 < a.C;
        float mu[2], is[2], k[2], m0[2], m1[2];
        for (int j = 0; j < 2; ++j) {
            const int c = c0 + j < a.C ? c0 + j : c0;
            mu[j] = a.mean[c]; is[j] = a.invstd[c]; k[j] = a.gamma[c] * is[j];
            m0[j] = (float)(a.acc[c] * invM); m1[j] = (float)(a.acc[a.C + c] * invM);
            if (blockIdx.x == 0 && rl == 0 && (j == 0 || has1)) { a.dbeta[c] = (float)a.acc[c]; a.dgamma[c] = (float)a.acc[a.C + c]; }
        }
        for (long r = (long)blockIdx.x * rowsPer + rl; r < a.M; r += (long)gridDim.x * rowsPer) {
            float2 g = load2(a.dz, (size_t)r * a.dz_ld + a.dz_off, c0, a.dz_cs, dvec && has1, has1);
            if (a.act) {
                const float2 zz = load2(a.z, (size_t)r * a.z_ld + a.z_off, c0, a.z_cs, zvec && has1, has1);
                g.x = act_grad(g.x, zz.x, a.act); g.y = act_grad(g.y, zz.y, a.act);
            }
            const float2 v = load2(a.y, (size_t)r * a.C, c0, 1, has1, has1);
            const float xh0 = (v.x - mu[0]) * is[0], xh1 = (v.y - mu[1]) * is[1];
            g.x = k[0] * (g.x - m0[0] - xh0 * m1[0]);
            g.y = k[1] * (g.y - m0[1] - xh1 * m1[1]);
            float* o = a.dy + (size_t)r * a.C + c0;
            if (has1) *reinterpret_cast<float2*>(o) = g; else o[0] = g.x;
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
    c.z = a.z; c.z_ld = a.z_ld; c.z_off = a.z_off; c.z_cs = a.z_cs;
    c.mean = a.mean; c.invstd = a.invstd; c.acc = a.acc; c.M = a.M; c.C = a.C; c.act = a.act; c.lanesC = L.lanesC;
    hipLaunchKernelGGL(col_reduce_kernel<2>, dim3(reduce_blocks(a.M, L.rowsPer)), dim3(256), 0, s, c);
    hipLaunchKernelGGL(bn_bwd_kernel, dim3(stream_blocks(a.M, L.rowsPer)), dim3(256), 0, s, a);
}

// =================================================================================================
// Weight gradient of a GEMM-shaped conv:  dW[n][k] += sum_m dY[m][n] * X[m][k]      (n < N = Cout, k < K)
//   pointwise: X[m][k] = x[m*ld + off + k];   dense 3x3: k = tap*Cin + ci, X = im2col of x (zero outside the image).
// Output is written in the reference's weight layout: pointwise [Cout][Cin]; dense [Cout][Cin][3][3].
// Block = 4 waves (2x2), output tile 64(n) x 64(k), one 32x32 f32 MFMA accumulator per wave; the M range of the
// block (grid.z slices) is consumed 32 rows at a time through LDS; float atomics add the slice into dW.
// =================================================================================================

__global__ __launch_bounds__(256) void wgrad_kernel(WgradArgs a)
{
    __shared__ float sdy[32][65];                // [m][n]
    __shared__ float sx[32][65];                 // [m][k]
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    const int wn = wave & 1, wk = wave >> 1;
    const int n0 = blockIdx.x * 64, k0 = blockIdx.y * 64;
    const int slices = gridDim.z;
    const int rows = ((a.M + slices - 1) / slices + 31) & ~31;
    const int m_begin = blockIdx.z * rows, m_end = min(a.M, m_begin + rows);
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int m0 = m_begin; m0 < m_end; m0 += 32) {
        // stage dY[32][64] and X[32][64]
        for (int i = t; i < 32 * 64; i += 256) {
            const int r = i >> 6, c = i & 63;
            const int m = m0 + r;
            float vy = 0.0f, vx = 0.0f;
            if (m < m_end) {
                if (n0 + c < a.N) vy = a.dy[(size_t)m * a.dy_ld + n0 + c];
                const int k = k0 + c;
                if (k < a.K) {
                    if (!a.dense) vx = a.x[(size_t)m * a.x_ld + a.x_off + k];
                    else {
                        const int tap = k / a.Cin, ci = k - tap * a.Cin;
                        const int hw = a.H * a.W, rem = m % hw;
                        const int y = rem / a.W + tap / 3 - 1, x = rem % a.W + tap % 3 - 1;
                        if (y >= 0 && y < a.H && x >= 0 && x < a.W)
                            vx = a.x[((size_t)m + (tap / 3 - 1) * a.W + (tap % 3 - 1)) * a.x_ld + a.x_off + ci];
                    }
                }
            }
            sdy[r][c] = vy;
            sx[r][c] = vx;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) {           // 2 rows (m) per MFMA step: lanes 0-31 take m = 2q, lanes 32-63 m = 2q+1
            const float av = sdy[2 * q + h][wn * 32 + l31];
            const float bv = sx[2 * q + h][wk * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
    // acc[r]: row i = (r&3) + 8*(r>>2) + 4*h  (n index), column j = l31 (k index)
    const int k = k0 + wk * 32 + l31;
    if (k >= a.K) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int n = n0 + wn * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        if (n >= a.N) continue;
        size_t idx;
        if (!a.dense) idx = (size_t)n * a.K + k;
        else { const int tap = k / a.Cin, ci = k - tap * a.Cin; idx = ((size_t)n * a.Cin + ci) * 9 + tap; }
        atomicAdd(a.dw + idx, acc[r]);
    }
}

void launch_wgrad(const WgradArgs& a, hipStream_t s)
{
    const int gn = (a.N + 63) / 64, gk = (a.K + 63) / 64;
    int slices = 1024 / (gn * gk);
    if (slices < 1) slices = 1;
    const int max_slices = (a.M + 127) / 128;
    if (slices > max_slices) slices = max_slices;
    if (slices < 1) slices = 1;
    hipLaunchKernelGGL(wgrad_kernel, dim3(gn, gk, slices), dim3(256), 0, s, a);
}

// ---- depthwise 3x3 weight gradient: dW[c][tap] += sum_p dY[p][c] * X[p*stride + tap - 1][c]  (torch layout [C][1][3][3])
//      Same lane layout as the column reductions: a lane owns two channels (float2 loads), row-lanes walk the output
//      pixels; 18 float accumulators per lane, LDS combine over the row-lanes, one float atomic per (c, tap) per block.
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int x_ld, int x_off,
                                                        int B, int H, int W, int C, int stride, float* __restrict__ dw, int lanesC)
{
    __shared__ float red[256][19];
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long Mo = (long)B * Ho * Wo;
    const int rowsPer = 256 / lanesC;
    const int cl = threadIdx.x & (lanesC - 1), rl = threadIdx.x / lanesC;
    const int CP = (C + 1) >> 1;
    const bool xvec = !(x_ld & 1) && !(x_off & 1), dvec = !(C & 1);
    for (int cp = cl; cp < ((CP + lanesC - 1) / lanesC) * lanesC; cp += lanesC) {
        const int c0 = cp * 2;
        const bool live = cp < CP, has1 = c0 + 1 < C;
        float acc[18];
#pragma unroll
        for (int k = 0; k < 18; ++k) acc[k] = 0.0f;
        if (live) {
            for (long p = (long)blockIdx.x * rowsPer + rl; p < Mo; p += (long)gridDim.x * rowsPer) {
                const int ox = (int)(p % Wo);
                const long q = p / Wo;
                const int oy = (int)(q % Ho), b = (int)(q / Ho);
                const float2 g = load2(dy, (size_t)p * C, c0, 1, dvec && has1, has1);
#pragma unroll
                for (int ky = 0; ky < 3; ++ky) {
                    const int iy = oy * stride - 1 + ky;
                    if (iy < 0 || iy >= H) continue;
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const int ix = ox * stride - 1 + kx;
                        if (ix < 0 || ix >= W) continue;
                        const float2 v = load2(x, ((size_t)(b * H + iy) * W + ix) * x_ld + x_off, c0, 1, xvec && has1, has1);
                        acc[(ky * 3 + kx) * 2 + 0] += g.x * v.x;
                        acc[(ky * 3 + kx) * 2 + 1] += g.y * v.y;
                    }
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 18; ++k) red[threadIdx.x][k] = acc[k];
        __syncthreads();
        if (rl == 0 && live) {
            for (int j = 1; j < rowsPer; ++j)
#pragma unroll
                for (int k = 0; k < 18; ++k) acc[k] += red[j * lanesC + cl][k];
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                atomicAdd(dw + (size_t)c0 * 9 + k, acc[2 * k]);
                if (has1) atomicAdd(dw + (size_t)(c0 + 1) * 9 + k, acc[2 * k + 1]);
            }
        }
    }
}

void launch_dw_wgrad(const float* dy, const float* x, int x_ld, int x_off, int B, int H, int W, int C, int stride, float* dw, hipStream_t s)
{
    const long Mo = (long)B * ((H - 1) / stride + 1) * ((W - 1) / stride + 1);
    const Lanes L = lanes_for(C);
    hipLaunchKernelGGL(dw_wgrad_kernel, dim3(reduce_blocks((int)Mo, L.rowsPer)), dim3(256), 0, s, dy, x, x_ld, x_off, B, H, W, C, stride, dw, L.lanesC);
}

// ---- depthwise 3x3 stride-2 input gradient: dX[iy][ix][c] = sum_{ky,kx} dY[(iy+1-ky)/2][(ix+1-kx)/2][c] * w[ky][kx][c]
//      over the taps for which the division is exact and the output pixel exists.  w packed [9][C].  accumulate: dX += .
__global__ __launch_bounds__(256) void dw_dgrad_s2_kernel(const float* __restrict__ dy, const float* __restrict__ w, int B, int H, int W, int C,
                                                           float* __restrict__ dx, int dx_ld, int dx_off, int accumulate)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * H * W * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int ix = (int)(p % W);
        const long q = p / W;
        const int iy = (int)(q % H), b = (int)(q / H);
        float acc = 0.0f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int ty = iy + 1 - ky;
            if (ty < 0 || (ty & 1)) continue;
            const int oy = ty >> 1;
            if (oy >= Ho) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int tx = ix + 1 - kx;
                if (tx < 0 || (tx & 1)) continue;
                const int ox = tx >> 1;
                if (ox >= Wo) continue;
                acc += dy[((size_t)(b * Ho + oy) * Wo + ox) * C + c] * w[(ky * 3 + kx) * C + c];
            }
        }
        float* d = dx + (size_t)p * dx_ld + dx_off + c;
        *d = accumulate ? *d + acc : acc;
    }
}

void launch_dw_dgrad_s2(const float* dy, const float* w, int B, int H, int W, int C, float* dx, int dx_ld, int dx_off, int accumulate, hipStream_t s)
{
    long blocks = ((long)B * H * W * C + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(dw_dgrad_s2_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dy, w, B, H, W, C, dx, dx_ld, dx_off, accumulate);
}

// ---- stem weight gradient: dW[co][ci][ky][kx] += sum_p dY[p][co] * x_nchw[b][ci][2oy-1+ky][2ox-1+kx]
//      As a GEMM: dW^T[r = ci*9+ky*3+kx (27 -> 32)][co (24 -> 32)] = Xpatch^T[32 x M] * dY[M x 32], one 32x32 f32 MFMA tile per
//      wave fed straight from global memory (lane l supplies patch element r = l%32 and dY column co = l%32 of output pixel
//      2*step + l/32).  A wave owns whole output rows; the four waves of a block are combined through LDS before the
//      float atomics into dW.
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int B, int H, int W, int Cout,
                                                          float* __restrict__ dw)
{
    __shared__ float red[3][64][17];
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
        const float* xrow = x + (((size_t)b * 3 + (rlive ? ci : 0)) * H + (yok ? iy : 0)) * W + (kx - 1);
        const float* drow = dy + (size_t)row * Wo * Cout + l31;
#pragma unroll 4
        for (int ox0 = 0; ox0 < Wo; ox0 += 2) {
            const int ox = ox0 + hh;
            const int ix = 2 * ox + kx - 1;
            float av = 0.0f, bv = 0.0f;
            if (yok && ox < Wo && ix >= 0 && ix < W) av = xrow[2 * ox];
            if (clive && ox < Wo) bv = drow[(size_t)ox * Cout];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    // acc[i]: row (patch element) = (i&3) + 8*(i>>2) + 4*hh, column (co) = l31
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) red[wave - 1][lane][i] = acc[i];
    }
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const float v = acc[i] + red[0][lane][i] + red[1][lane][i] + red[2][lane][i];
            const int rr = (i & 3) + 8 * (i >> 2) + 4 * hh;
            if (rr < 27 && clive) atomicAdd(dw + (size_t)l31 * 27 + rr, v);
        }
    }
}

void launch_stem_wgrad(const float* dy, const float* x, int B, int H, int W, int Cout, float* dw, hipStream_t s)
{
    const int nrows = B * ((H - 1) / 2 + 1);
    int G = (nrows + 3) / 4;
    if (G > 512) G = 512;
    if (G < 1) G = 1;
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(G), dim3(256), 0, s, dy, x, B, H, W, Cout, dw);
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

__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dy, const int32_t* __restrict__ idx, int B, int H, int W, int C, float* __restrict__ dx)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * Wo * C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % C);
        const long p = i / C;
        const int b = (int)(p / ((long)Ho * Wo));
        atomicAdd(dx + ((size_t)b * H * W + idx[i]) * C + c, dy[i]);
    }
}

void launch_maxpool_idx(const float* x, int B, int H, int W, int C, float* y, int32_t* idx, hipStream_t s)
{
    long blocks = ((long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * C + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(maxpool_idx_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, B, H, W, C, y, idx);
}

void launch_maxpool_bwd(const float* dy, const int32_t* idx, int B, int H, int W, int C, float* dx, hipStream_t s)
{
    long blocks = ((long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1) * C + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    hipLaunchKernelGGL(maxpool_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dy, idx, B, H, W, C, dx);
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

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
// Column reductions over an [M, C] matrix (row stride ld, channel offset off), two-stage and deterministic:
//   MODE 0: s0 = sum x                                  (BN mean, bias gradient)
//   MODE 1: s0 = sum (x - mean)^2                       (BN variance, second pass)
//   MODE 2: s0 = sum dyh, s1 = sum dyh * xhat           (BN backward; dyh = dz * act'(z), xhat = (y-mean)*invstd)
// partial[g][2][C]; col_finalize turns them into what the consumer needs.
// =================================================================================================
struct ColArgs {
    const float* x; int x_ld, x_off;            // MODE 0/1: the matrix; MODE 2: y (pre-BN conv output)
    const float* dz; int dz_ld, dz_off, dz_cs;  // MODE 2: upstream gradient (may be a strided channel view)
    const float* z; int z_ld, z_off, z_cs;      // MODE 2: BN+act output (for act')
    const float* mean; const float* invstd;
    float* partial; int M, C, act;
};

template <int MODE>
__global__ __launch_bounds__(256) void col_reduce_kernel(ColArgs a)
{
    const int G = gridDim.x;
    const int rows = (a.M + G - 1) / G;
    const int r0 = blockIdx.x * rows, r1 = min(a.M, r0 + rows);
    for (int c = threadIdx.x; c < a.C; c += 256) {
        float s0 = 0.0f, s1 = 0.0f;
        float mu = 0.0f, is = 0.0f;
        if (MODE >= 1) mu = a.mean[c];
        if (MODE == 2) is = a.invstd[c];
        for (int r = r0; r < r1; ++r) {
            if (MODE == 0) s0 += a.x[(size_t)r * a.x_ld + a.x_off + c];
            else if (MODE == 1) { const float d = a.x[(size_t)r * a.x_ld + a.x_off + c] - mu; s0 += d * d; }
            else {
                float g = a.dz[(size_t)r * a.dz_ld + a.dz_off + c * a.dz_cs];
                if (a.act) {
                    const float zz = a.z[(size_t)r * a.z_ld + a.z_off + c * a.z_cs];
                    g = zz > 0.0f ? g : (a.act == 2 ? 0.1f * g : 0.0f);
                }
                const float xh = (a.x[(size_t)r * a.x_ld + a.x_off + c] - mu) * is;
                s0 += g; s1 += g * xh;
            }
        }
        a.partial[((size_t)blockIdx.x * 2 + 0) * a.C + c] = s0;
        a.partial[((size_t)blockIdx.x * 2 + 1) * a.C + c] = s1;
    }
}

// what: 0 -> out0 = s0 / M (mean) ; 1 -> out0 = 1/sqrt(s0/M + eps) (invstd), out1 = s0/M (biased var);
//       2 -> out0 += s0 (dbeta), out1 += s1 (dgamma), and out2/out3 = s0/M, s1/M for the dx pass ; 3 -> out0 += s0 (bias grad)
__global__ __launch_bounds__(256) void col_finalize_kernel(const float* __restrict__ partial, int G, int C, int M, float eps, int what,
                                                            float* out0, float* out1, float* out2, float* out3)
{
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= C) return;
    double s0 = 0.0, s1 = 0.0;
    for (int g = 0; g < G; ++g) { s0 += (double)partial[((size_t)g * 2 + 0) * C + c]; s1 += (double)partial[((size_t)g * 2 + 1) * C + c]; }
    if (what == 0) out0[c] = (float)(s0 / M);
    else if (what == 1) { const double var = s0 / M; out0[c] = (float)(1.0 / sqrt(var + (double)eps)); out1[c] = (float)var; }
    else if (what == 2) { out0[c] += (float)s0; out1[c] += (float)s1; out2[c] = (float)(s0 / M); out3[c] = (float)(s1 / M); }
    else out0[c] += (float)s0;
}

static int col_groups(int M) { int g = (M + 255) / 256; if (g > 512) g = 512; if (g < 1) g = 1; return g; }

int col_partial_floats(int M, int C) { return col_groups(M) * 2 * C; }

void launch_col_stats(const float* y, int ld, int off, int M, int C, float eps, float* partial, float* mean, float* invstd, float* var, hipStream_t s)
{
    ColArgs a{};
    a.x = y; a.x_ld = ld; a.x_off = off; a.partial = partial; a.M = M; a.C = C; a.mean = mean;
    const int G = col_groups(M);
    hipLaunchKernelGGL(col_reduce_kernel<0>, dim3(G), dim3(256), 0, s, a);
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, partial, G, C, M, eps, 0, mean, (float*)nullptr, (float*)nullptr, (float*)nullptr);
    hipLaunchKernelGGL(col_reduce_kernel<1>, dim3(G), dim3(256), 0, s, a);
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, partial, G, C, M, eps, 1, invstd, var, (float*)nullptr, (float*)nullptr);
}

void launch_col_sum_accumulate(const float* x, int ld, int off, int M, int C, float* partial, float* out, hipStream_t s)
{
    ColArgs a{};
    a.x = x; a.x_ld = ld; a.x_off = off; a.partial = partial; a.M = M; a.C = C;
    const int G = col_groups(M);
    hipLaunchKernelGGL(col_reduce_kernel<0>, dim3(G), dim3(256), 0, s, a);
    hipLaunchKernelGGL(col_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, s, partial, G, C, M, 0.0f, 3, out, (float*)nullptr, (float*)nullptr, (float*)nullptr);
}

// ---- BatchNorm forward apply: z = act((y - mean) * invstd * gamma + beta); optional channel-interleaved output
//      (out[m][off + c*cs]) and pass-through copy (out[m][pass_dst_off + c*cs] = pass[m][pass_off + c]) = concat+shuffle;
//      also the running-statistics update (momentum 0.1, unbiased variance) by block 0.

__global__ __launch_bounds__(256) void bn_apply_kernel(BnApplyArgs a)
{
    const long total = (long)a.M * a.C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % a.C);
        const long m = i / a.C;
        float v = (a.y[i] - a.mean[c]) * a.invstd[c] * a.gamma[c] + a.beta[c];
        if (a.act == 1) v = v > 0.0f ? v : 0.0f;
        else if (a.act == 2) v = v > 0.0f ? v : 0.1f * v;
        a.out[(size_t)m * a.out_ld + a.out_off + c * a.out_cs] = v;
        if (a.pass) a.out[(size_t)m * a.out_ld + a.pass_dst_off + c * a.out_cs] = a.pass[(size_t)m * a.pass_ld + a.pass_off + c];
    }
    if (blockIdx.x == 0 && a.rmean) {
        for (int c = threadIdx.x; c < a.C; c += 256) {
            const float unbiased = a.M > 1 ? a.var[c] * ((float)a.M / (float)(a.M - 1)) : a.var[c];
            a.rmean[c] = (1.0f - a.momentum) * a.rmean[c] + a.momentum * a.mean[c];
            a.rvar[c] = (1.0f - a.momentum) * a.rvar[c] + a.momentum * unbiased;
        }
    }
}

void launch_bn_apply(const BnApplyArgs& a, hipStream_t s)
{
    long blocks = ((long)a.M * a.C + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bn_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
}

// ---- BatchNorm backward: dy = gamma * invstd * (dyh - mean(dyh) - xhat * mean(dyh * xhat)),  dyh = dz * act'(z).
//      With gamma == nullptr it is the plain activation backward (layers without BN).

__global__ __launch_bounds__(256) void bn_bwd_kernel(BnBwdArgs a)
{
    const long total = (long)a.M * a.C;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int c = (int)(i % a.C);
        const long m = i / a.C;
        float g = a.dz[(size_t)m * a.dz_ld + a.dz_off + c * a.dz_cs];
        if (a.act) {
            const float zz = a.z[(size_t)m * a.z_ld + a.z_off + c * a.z_cs];
            g = zz > 0.0f ? g : (a.act == 2 ? 0.1f * g : 0.0f);
        }
        if (a.gamma) {
            const float xh = (a.y[i] - a.mean[c]) * a.invstd[c];
            g = a.gamma[c] * a.invstd[c] * (g - a.m_dyh[c] - xh * a.m_dyhx[c]);
        }
        a.dy[i] = g;
    }
}

void launch_bn_bwd(const BnBwdArgs& a, float* partial, float* dgamma, float* dbeta, float* scratch2C, hipStream_t s)
{
    BnBwdArgs b = a;
    if (a.gamma) {
        ColArgs c{};
        c.x = a.y; c.x_ld = a.C; c.x_off = 0;
        c.dz = a.dz; c.dz_ld = a.dz_ld; c.dz_off = a.dz_off; c.dz_cs = a.dz_cs;
        c.z = a.z; c.z_ld = a.z_ld; c.z_off = a.z_off; c.z_cs = a.z_cs;
        c.mean = a.mean; c.invstd = a.invstd; c.partial = partial; c.M = a.M; c.C = a.C; c.act = a.act;
        const int G = col_groups(a.M);
        hipLaunchKernelGGL(col_reduce_kernel<2>, dim3(G), dim3(256), 0, s, c);
        hipLaunchKernelGGL(col_finalize_kernel, dim3((a.C + 255) / 256), dim3(256), 0, s, partial, G, a.C, a.M, 0.0f, 2, dbeta, dgamma, scratch2C, scratch2C + a.C);
        b.m_dyh = scratch2C; b.m_dyhx = scratch2C + a.C;
    }
    long blocks = ((long)a.M * a.C + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(bn_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, s, b);
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
__global__ __launch_bounds__(256) void dw_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int x_ld, int x_off,
                                                        int B, int H, int W, int C, int stride, float* __restrict__ dw)
{
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;
    const long Mo = (long)B * Ho * Wo;
    const long rows = (Mo + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * rows, p1 = min(Mo, p0 + rows);
    for (int c = threadIdx.x; c < C; c += 256) {
        float acc[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) acc[k] = 0.0f;
        for (long p = p0; p < p1; ++p) {
            const int ox = (int)(p % Wo);
            const long q = p / Wo;
            const int oy = (int)(q % Ho), b = (int)(q / Ho);
            const float g = dy[(size_t)p * C + c];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int iy = oy * stride - 1 + ky;
                if (iy < 0 || iy >= H) continue;
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int ix = ox * stride - 1 + kx;
                    if (ix < 0 || ix >= W) continue;
                    acc[ky * 3 + kx] += g * x[((size_t)(b * H + iy) * W + ix) * x_ld + x_off + c];
                }
            }
        }
#pragma unroll
        for (int k = 0; k < 9; ++k) atomicAdd(dw + (size_t)c * 9 + k, acc[k]);
    }
}

void launch_dw_wgrad(const float* dy, const float* x, int x_ld, int x_off, int B, int H, int W, int C, int stride, float* dw, hipStream_t s)
{
    const long Mo = (long)B * ((H - 1) / stride + 1) * ((W - 1) / stride + 1);
    int G = (int)((Mo + 127) / 128);
    if (G > 1024) G = 1024;
    if (G < 1) G = 1;
    hipLaunchKernelGGL(dw_wgrad_kernel, dim3(G), dim3(256), 0, s, dy, x, x_ld, x_off, B, H, W, C, stride, dw);
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
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ dy, const float* __restrict__ x, int B, int H, int W, int Cout,
                                                          float* __restrict__ dw)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long Mo = (long)B * Ho * Wo;
    const long rows = (Mo + gridDim.x - 1) / gridDim.x;
    const long p0 = (long)blockIdx.x * rows, p1 = min(Mo, p0 + rows);
    const int nout = Cout * 27;
    for (int o = threadIdx.x; o < nout; o += 256) {
        const int co = o / 27, r = o - co * 27;
        const int ci = r / 9, ky = (r % 9) / 3, kx = r % 3;
        float acc = 0.0f;
        for (long p = p0; p < p1; ++p) {
            const int ox = (int)(p % Wo);
            const long q = p / Wo;
            const int oy = (int)(q % Ho), b = (int)(q / Ho);
            const int iy = oy * 2 - 1 + ky, ix = ox * 2 - 1 + kx;
            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
            acc += dy[(size_t)p * Cout + co] * x[(((size_t)b * 3 + ci) * H + iy) * W + ix];
        }
        atomicAdd(dw + o, acc);
    }
}

void launch_stem_wgrad(const float* dy, const float* x, int B, int H, int W, int Cout, float* dw, hipStream_t s)
{
    const long Mo = (long)B * ((H - 1) / 2 + 1) * ((W - 1) / 2 + 1);
    int G = (int)((Mo + 255) / 256);
    if (G > 2048) G = 2048;
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

// kernels_conv.hip — gfx950 convolution kernels of the YOLO-Nano hot path (float32, NHWC).
//
//   gemm_conv_kernel   pointwise 1x1 and dense 3x3 convolutions as GEMMs on the f32 MFMA
//                      (v_mfma_f32_32x32x2_f32), LDS double-buffered, fused bias + activation,
//                      fused FPN/PAN resample-add prologue (3x3), fused concat+channel-shuffle
//                      epilogue (pointwise).
//   dwconv3x3_kernel   depthwise 3x3 (stride 1/2), one thread per (pixel, channel pair)
//   stem_kernel        3->24 3x3 stride-2 conv reading NCHW input, writing NHWC
//   maxpool_kernel     3x3 stride-2 max pool
//
// Reference semantics: backbone/shufflenetv2.py:31-78,109-116, utils/modules.py:8-18,
// models/yolo_nano.py:286-301 — with BatchNorm folded into the weights (utils/fuse_conv_bn.py:6-22).
#include "yn_internal.h"

namespace ynk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

static thread_local const char* g_last_kernel = "";
const char* last_kernel_name() { return g_last_kernel; }
void set_last_kernel_name(const char* n) { g_last_kernel = n; }

__device__ __forceinline__ float apply_act(float v, int act)
{
    if (act == 1) return v > 0.0f ? v : 0.0f;
    if (act == 2) return v > 0.0f ? v : 0.1f * v;
    return v;
}

// -------------------------------------------------------------------------------------------------
// GEMM convolution.  Block = 4 waves laid out WM x WN; each wave owns a 32 x (32*NT) output tile and
// keeps it in NT 32x32 f32 MFMA accumulators.  K is consumed in chunks of 32 through two LDS buffers:
//   As[kp][row] float2  (k-pair major; +1 float2 pad per kp row => conflict-free ds_write_b64)
//   Bs[kp][n]   float2  (straight copy of the packed weights)
// One ds_read_b64 of A and of B per lane feeds two MFMAs: lanes 0-31 carry k = 4q, 4q+1 and lanes
// 32-63 carry k = 4q+2, 4q+3 (the order of the k-sum inside a chunk is free as long as A and B agree).
// -------------------------------------------------------------------------------------------------
template <int WM, int WN, int NT, int MODE>
__global__ __launch_bounds__(256) void gemm_conv_kernel(GemmArgs a)
{
    constexpr int BM = 32 * WM, BN = 32 * NT * WN, KP = 16;
    constexpr int AS = BM * 2 + 2;              // floats per kp row of A
    constexpr int BS = BN * 2;                  // floats per kp row of B
    constexpr int A_PER = BM / 16;              // float2 per thread per chunk
    constexpr int B_PER = BN / 32;              // float4 per thread per chunk
    __shared__ __attribute__((aligned(16))) float smem[2 * KP * (AS + BS)];
    float* As = smem;
    float* Bs = smem + 2 * KP * AS;

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    const int m0 = blockIdx.x * BM;
    const int n0 = blockIdx.y * BN;

    const int Ktot = (MODE == 1) ? 9 * a.K : a.K;          // MODE 1: a.K = Cin
    const int nchunks = (Ktot + 31) >> 5;
    const int kp_total = (Ktot + 1) >> 1;

    // ---- per-thread A row bookkeeping ----
    const int a_kp = t & 15;
    int a_m[A_PER];
    int a_yx[A_PER];                                       // MODE 1: (y << 16) | x, or -1 if row invalid
#pragma unroll
    for (int i = 0; i < A_PER; ++i) {
        const int r = (t >> 4) + 16 * i;
        const int m = m0 + r;
        a_m[i] = (m < a.M) ? m : -1;
        if (MODE == 1) {
            if (m < a.M) {
                const int hw = a.H * a.W;
                const int rem = m % hw;
                a_yx[i] = ((rem / a.W) << 16) | (rem % a.W);
            } else a_yx[i] = 0;
        }
    }

    float2 a_reg[A_PER];
    float4 b_reg[B_PER];

    auto prefetch = [&](int c) {
        const int k0 = c << 5;
        if (MODE == 0) {
            const int k = k0 + 2 * a_kp;
            const bool kv = k < a.K;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                float2 v = make_float2(0.0f, 0.0f);
                if (kv && a_m[i] >= 0) v = *reinterpret_cast<const float2*>(a.in + (size_t)a_m[i] * a.in_ld + a.in_off + k);
                a_reg[i] = v;
            }
        } else {
            const int cpt = a.K >> 5;                       // chunks per tap
            const int tap = c / cpt;
            const int ci = ((c - tap * cpt) << 5) + 2 * a_kp;
            const int dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
            for (int i = 0; i < A_PER; ++i) {
                float2 v = make_float2(0.0f, 0.0f);
                if (a_m[i] >= 0) {
                    const int y = (a_yx[i] >> 16) + dy, x = (a_yx[i] & 0xffff) + dx;
                    if (y >= 0 && y < a.H && x >= 0 && x < a.W) {
                        const int src = a_m[i] + dy * a.W + dx;
                        v = *reinterpret_cast<const float2*>(a.in + (size_t)src * a.in_ld + a.in_off + ci);
                        if (a.resample) {
                            const int b = a_m[i] / (a.H * a.W);
                            size_t p2;
                            if (a.resample == 1) p2 = ((size_t)b * (a.H >> 1) + (y >> 1)) * (a.W >> 1) + (x >> 1);
                            else                 p2 = ((size_t)b * (a.H << 1) + (y << 1)) * (a.W << 1) + (x << 1);
                            const float2 u = *reinterpret_cast<const float2*>(a.in2 + p2 * a.K + ci);
                            v.x += u.x; v.y += u.y;
                        }
                    }
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
            float4 v = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
            if (kpg < kp_total && n < a.Npad) v = *reinterpret_cast<const float4*>(a.Wp + ((size_t)kpg * a.Npad + n) * 2);
            b_reg[i] = v;
        }
    };
    auto stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            const int r = (t >> 4) + 16 * i;
            *reinterpret_cast<float2*>(As + buf * KP * AS + a_kp * AS + r * 2) = a_reg[i];
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = t + 256 * i;
            const int kp = idx / (BN / 2), c4 = idx - kp * (BN / 2);
            *reinterpret_cast<float4*>(Bs + buf * KP * BS + kp * BS + c4 * 4) = b_reg[i];
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
        const int buf = c & 1;
        if (c + 1 < nchunks) prefetch(c + 1);
        const int krem = Ktot - (c << 5);
        const int nq = krem >= 32 ? 8 : ((krem + 3) >> 2);
        const float* Ab = As + buf * KP * AS + (wm * 32 + l31) * 2;
        const float* Bb = Bs + buf * KP * BS + (wn * NT * 32 + l31) * 2;
        for (int q = 0; q < nq; ++q) {
            const float2 av = *reinterpret_cast<const float2*>(Ab + (2 * q + h) * AS);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const float2 bv = *reinterpret_cast<const float2*>(Bb + (2 * q + h) * BS + nt * 64);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc[nt], 0, 0, 0);
                acc[nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc[nt], 0, 0, 0);
            }
        }
        if (c + 1 < nchunks) stage(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias + activation (+ concat/shuffle interleave with the pass-through half) ----
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = n0 + (wn * NT + nt) * 32 + l31;
        if (n >= a.N) continue;
        const float bias = a.bias[n];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
            const int m = m0 + wm * 32 + row;
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

struct TileCfg { int WM, WN, NT; };

static TileCfg choose_tile(int M, int Npad)
{
    static const TileCfg cands[] = {{4, 1, 4}, {4, 1, 3}, {4, 1, 2}, {4, 1, 1}, {2, 2, 2}, {2, 2, 1}, {1, 4, 2}, {1, 4, 1}};
    TileCfg best = cands[0];
    double best_cost = 1e30;
    for (const TileCfg& c : cands) {
        const int BM = 32 * c.WM, BN = 32 * c.NT * c.WN;
        const long gx = (M + BM - 1) / BM, gy = (Npad + BN - 1) / BN;
        const long blocks = gx * gy;
        const long rounds = (blocks + 511) / 512;          // ~2 resident blocks on each of 256 CUs
        const double cost = (double)rounds * BM * BN * (1.0 + 0.02 * (128 / BM - 1));
        if (cost < best_cost) { best_cost = cost; best = c; }
    }
    return best;
}

template <int MODE>
static void launch_gemm(const GemmArgs& a, hipStream_t s)
{
    const TileCfg c = choose_tile(a.M, a.Npad);
    const int BM = 32 * c.WM, BN = 32 * c.NT * c.WN;
    dim3 grid((a.M + BM - 1) / BM, (a.Npad + BN - 1) / BN);
#define YN_GEMM_CASE(wm, wn, nt)                                                                     \
    if (c.WM == wm && c.WN == wn && c.NT == nt) {                                                    \
        g_last_kernel = MODE ? "gemm_conv_kernel<" #wm "," #wn "," #nt ",1>" : "gemm_conv_kernel<" #wm "," #wn "," #nt ",0>"; \
        hipLaunchKernelGGL((gemm_conv_kernel<wm, wn, nt, MODE>), grid, dim3(256), 0, s, a);          \
        return;                                                                                      \
    }
    YN_GEMM_CASE(4, 1, 4) YN_GEMM_CASE(4, 1, 3) YN_GEMM_CASE(4, 1, 2) YN_GEMM_CASE(4, 1, 1)
    YN_GEMM_CASE(2, 2, 2) YN_GEMM_CASE(2, 2, 1) YN_GEMM_CASE(1, 4, 2) YN_GEMM_CASE(1, 4, 1)
#undef YN_GEMM_CASE
}

void launch_pw(const GemmArgs& a, hipStream_t s) { launch_gemm<0>(a, s); }
void launch_conv3x3(const GemmArgs& a, hipStream_t s) { launch_gemm<1>(a, s); }

// -------------------------------------------------------------------------------------------------
// Depthwise 3x3, pad 1, stride 1 or 2.  Thread = (output pixel, channel pair); consecutive threads walk
// the channel pairs of one pixel and then the next pixel, so loads and stores are fully coalesced in
// NHWC.  The nine taps of neighbouring pixels overlap in L1/L2.
// -------------------------------------------------------------------------------------------------
template <int STRIDE>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(DwArgs a)
{
    const int Ho = (a.H - 1) / STRIDE + 1, Wo = (a.W - 1) / STRIDE + 1;
    const int cp_n = a.C >> 1;
    const long total = (long)a.B * Ho * Wo * cp_n;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int cp = (int)(i % cp_n);
        const long p = i / cp_n;
        const int ox = (int)(p % Wo);
        const long q = p / Wo;
        const int oy = (int)(q % Ho);
        const int b = (int)(q / Ho);
        const int c = cp * 2;
        float2 acc = *reinterpret_cast<const float2*>(a.bias + c);
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * STRIDE - 1 + ky;
            if (iy < 0 || iy >= a.H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * STRIDE - 1 + kx;
                if (ix < 0 || ix >= a.W) continue;
                const float2 v = *reinterpret_cast<const float2*>(a.in + ((size_t)(b * a.H + iy) * a.W + ix) * a.in_ld + a.in_off + c);
                const float2 w = *reinterpret_cast<const float2*>(a.w + (ky * 3 + kx) * a.C + c);
                acc.x += v.x * w.x;
                acc.y += v.y * w.y;
            }
        }
        acc.x = apply_act(acc.x, a.act);
        acc.y = apply_act(acc.y, a.act);
        *reinterpret_cast<float2*>(a.out + (size_t)p * a.out_ld + a.out_off + c) = acc;
    }
}

void launch_dw(const DwArgs& a, hipStream_t s)
{
    const int Ho = (a.H - 1) / a.stride + 1, Wo = (a.W - 1) / a.stride + 1;
    const long total = (long)a.B * Ho * Wo * (a.C >> 1);
    long blocks = (total + 255) / 256;
    if (blocks > 256 * 32) blocks = 256 * 32;
    if (blocks < 1) blocks = 1;
    g_last_kernel = a.stride == 1 ? "dwconv3x3_kernel<1>" : "dwconv3x3_kernel<2>";
    if (a.stride == 1) hipLaunchKernelGGL(dwconv3x3_kernel<1>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    else               hipLaunchKernelGGL(dwconv3x3_kernel<2>, dim3((unsigned)blocks), dim3(256), 0, s, a);
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
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * Wo;
    const long p = (long)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int ox = (int)(p % Wo);
    const long q = p / Wo;
    const int oy = (int)(q % Ho);
    const int b = (int)(q / Ho);
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = bias[co];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
        const float* xp = x + ((size_t)b * 3 + ci) * H * W;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int iy = oy * 2 - 1 + ky;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int ix = ox * 2 - 1 + kx;
                float v = 0.0f;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = xp[(size_t)iy * W + ix];
                const float* wr = w + ((ci * 3 + ky) * 3 + kx) * COUT;
#pragma unroll
                for (int co = 0; co < COUT; ++co) acc[co] += v * wr[co];
            }
        }
    }
    float* yo = y + (size_t)p * COUT;
#pragma unroll
    for (int co = 0; co < COUT; co += 4) {
        float4 o = make_float4(apply_act(acc[co], act), apply_act(acc[co + 1], act), apply_act(acc[co + 2], act), apply_act(acc[co + 3], act));
        *reinterpret_cast<float4*>(yo + co) = o;
    }
}

void launch_stem(const float* x, int B, int H, int W, const float* w, const float* bias, int Cout, int act, float* y, hipStream_t s)
{
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)B * Ho * Wo;
    const unsigned blocks = (unsigned)((total + 255) / 256);
    g_last_kernel = "stem_kernel<24>";
    if (Cout == 24) hipLaunchKernelGGL(stem_kernel<24>, dim3(blocks), dim3(256), 0, s, x, B, H, W, w, bias, act, y);
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

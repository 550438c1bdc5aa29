// kernels_unit.hip — one kernel per stride-1 ShuffleV2 unit (backbone/shufflenetv2.py:53-63, 70-72, 14-28):
//
//     x = [x1 | x2]  ->  out = shuffle(cat(x1, pw2(dw3x3(pw1(x2)))))        (BatchNorm folded, ReLU after pw1 / pw2)
//
// The three-kernel version (pointwise GEMM, depthwise, pointwise GEMM with the concat+shuffle epilogue) moves the two
// intermediate tensors through memory and, for the 26x26 / 13x13 maps of stages 3-4, spends most of each launch on
// latency (a few hundred small blocks).  Here a block owns a TH x TW spatial tile of ONE image:
//   1. the (TH+2) x (TW+2) halo of x2 is staged in LDS (unconditional clamped loads, zero outside the image);
//   2. pw1 runs on every halo pixel (f32 MFMA, rows = halo pixels, all N per wave, W1 streamed through a double
//      buffer) and its ReLU output OVERWRITES the halo tile in place — a wave only ever reads the A rows of its own row
//      tiles; pixels outside the image are written as 0 (they are the depthwise conv's zero padding);
//   3. pw2 runs on the TH x TW interior; its A fragment (pixel, two channels) is the depthwise 3x3 evaluated on the fly
//      from the LDS tile (9 ds_read_b64 of activations + 9 of taps per k-step, shared by all N tiles of the wave);
//   4. the epilogue interleaves the pass-through half: out[2j] = x1[j], out[2j+1] = relu(pw2[j]).
// HBM traffic per unit: x2 (with ~1.3x halo overlap) + x1 in, out written once — about half of the three-kernel version.
#include "yn_internal.h"
#include "yn_device.h"

namespace ynk {

// the opaque-mask load idiom of yn_device.h under this file's historical names
__device__ __forceinline__ unsigned unit_mask(bool ok) { return opaque_mask(ok); }
__device__ __forceinline__ float unit_keep(float v, unsigned mk) { return __uint_as_float(__float_as_uint(v) & mk); }

template <int NT, int KP>
__global__ __launch_bounds__(256) void shuffle_unit_kernel(UnitArgs a)
{
    constexpr int BN = 32 * NT, BS = BN * 2, B_PER = KP * BN / 512, NQC = KP / 2;   // NQC k-steps (of 4 k) per weight chunk
    extern __shared__ __attribute__((aligned(16))) float un_smem[];
    const int bf = a.bf, C = 2 * bf, CS = a.CS;
    const int HWd = a.TW + 2, HP = (a.TH + 2) * HWd, OP = a.TH * a.TW;
    const int RT1 = (HP + 31) >> 5, RT2 = (OP + 31) >> 5;
    float* tile = un_smem;                                   // [RT1*32][CS]
    float* Bs = tile + (size_t)RT1 * 32 * CS;                // [2][KP][BS]
    float* wd = Bs + 2 * KP * BS;                            // [10][CS] (+8 zero floats): 9 taps, bias; pad columns zero

    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, l31 = lane & 31, h = lane >> 5;
    int blk = blockIdx.x;
    const int tx = blk % a.tilesX; blk /= a.tilesX;
    const int ty = blk % a.tilesY;
    const int b = blk / a.tilesY;
    const int y0 = ty * a.TH, x0 = tx * a.TW;
    const int th = min(a.TH, a.H - y0), tw = min(a.TW, a.W - x0);
    const float* xb = a.x + (size_t)b * a.H * a.W * C;
    float* ob = a.out + (size_t)b * a.H * a.W * C;
    const int nchunks = (bf + 2 * KP - 1) / (2 * KP), kp_total = (bf + 1) >> 1;
    // Every chunk runs all NQC k-steps: weight rows past K are zero (masked loads) and the activations they meet are finite
    // (zeroed pad columns, or the first floats of the next LDS row), so the k loops have no data-dependent control flow.

    float4 b_reg[B_PER];
    auto prefetch_b = [&](const float* Wp, int c) {
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int idx = t + 256 * i;
            const int kp = idx / (BN / 2), c4 = idx - kp * (BN / 2);
            const int kpg = c * KP + kp, n = c4 * 2;
            const bool ok = kpg < kp_total && n < a.Npad;
            const float4 v = *reinterpret_cast<const float4*>(Wp + ((size_t)(ok ? kpg : 0) * a.Npad + (ok ? n : 0)) * 2);
            const unsigned mk = unit_mask(ok);
            b_reg[i] = make_float4(unit_keep(v.x, mk), unit_keep(v.y, mk), unit_keep(v.z, mk), unit_keep(v.w, mk));
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
#ifdef YN_EXP_TIMING
    const long long T0 = __builtin_readcyclecounter();
#endif
    prefetch_b(a.Wp1, 0);

    // ---- 1. depthwise taps + bias, halo of x2 -> LDS -------------------------------------------------------
    for (int i = t; i < 10 * CS + 8; i += 256) {
        const int row = i / CS, c = i - row * CS;
        wd[i] = (row < 10 && c < bf) ? (row < 9 ? a.wdw[row * bf + c] : a.bdw[c]) : 0.0f;
    }
    for (int i = t; i < RT1 * 32 * (CS - bf); i += 256) {    // pad columns of the tile: read by the last (partial) k-step
        const int row = i / (CS - bf), c = i - row * (CS - bf);
        tile[(size_t)row * CS + bf + c] = 0.0f;
    }
    {
        // thread = (channel group of V floats, pixel lane); V = 4 when bf % 4 == 0 (16-byte loads), else 2
        constexpr int U = 32;                                // the whole tile in ONE batch of loads (8 pixel lanes x 32 = 256 rows)
        const int V = (bf & 3) ? 2 : 4;
        const int cgn = bf / V;
        const int ppl = 256 / cgn;
        const int cg = t % cgn, pl = t / cgn;
        const int rows = RT1 * 32;
        if (pl < ppl) {
            for (int i0 = pl; i0 < rows; i0 += ppl * U) {
                float4 v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int hp = i0 + u * ppl;
                    const int hy = hp / HWd, hx = hp - hy * HWd;
                    const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
                    const bool ok = hp < HP && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W;
                    const int yc = min(max(yy, 0), a.H - 1), xc = min(max(xx, 0), a.W - 1);
                    const float* src = xb + ((size_t)yc * a.W + xc) * C + bf + V * cg;
                    const unsigned mk = unit_mask(ok);
                    if (V == 4) {
                        const float4 g = *reinterpret_cast<const float4*>(src);
                        v[u] = make_float4(unit_keep(g.x, mk), unit_keep(g.y, mk), unit_keep(g.z, mk), unit_keep(g.w, mk));
                    } else {
                        const float2 g = *reinterpret_cast<const float2*>(src);
                        v[u] = make_float4(unit_keep(g.x, mk), unit_keep(g.y, mk), 0.0f, 0.0f);
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int hp = i0 + u * ppl;
                    if (hp < rows) {                         // pixel stride CS*4 bytes is only 8-byte aligned
                        *reinterpret_cast<float2*>(tile + (size_t)hp * CS + V * cg) = make_float2(v[u].x, v[u].y);
                        if (V == 4) *reinterpret_cast<float2*>(tile + (size_t)hp * CS + V * cg + 2) = make_float2(v[u].z, v[u].w);
                    }
                }
            }
        }
    }
    stage_b(0);
    __syncthreads();

#ifdef YN_EXP_TIMING
    const long long T1 = __builtin_readcyclecounter();
#endif
    // ---- 2. pw1 on the halo pixels, ReLU, in place -----------------------------------------------------------
    f32x16 acc[2][NT];
    auto zero_acc = [&]() {
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2)
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[r2][i][k] = 0.0f;
    };
    zero_acc();
    {
        const bool has1 = wave + 4 < RT1;
        const float* A0 = tile + (size_t)(wave * 32 + l31) * CS + 2 * h;
        const float* A1 = tile + (size_t)((has1 ? wave + 4 : wave) * 32 + l31) * CS + 2 * h;
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunks) prefetch_b(a.Wp1, c + 1);
            const float* Bb = Bs + buf * KP * BS + l31 * 2 + h * BS;
            const int kc = c * 2 * KP;
            float2 a0 = *reinterpret_cast<const float2*>(A0 + kc), a1 = *reinterpret_cast<const float2*>(A1 + kc);
            float2 bv[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const float2*>(Bb + nt * 64);
#pragma unroll
            for (int q = 0; q < NQC; ++q) {
                float2 a0n = a0, a1n = a1, bvn[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bvn[nt] = bv[nt];
                if (q + 1 < NQC) {                           // fragments of step q+1 are requested before the MFMAs of step q
                    a0n = *reinterpret_cast<const float2*>(A0 + kc + 4 * (q + 1));
                    a1n = *reinterpret_cast<const float2*>(A1 + kc + 4 * (q + 1));
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bvn[nt] = *reinterpret_cast<const float2*>(Bb + 2 * (q + 1) * BS + nt * 64);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bv[nt].x, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bv[nt].x, acc[1][nt], 0, 0, 0);
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bv[nt].y, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bv[nt].y, acc[1][nt], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                a0 = a0n; a1 = a1n;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = bvn[nt];
            }
            if (c + 1 < nchunks) { stage_b(buf ^ 1); __syncthreads(); }
        }
        prefetch_b(a.Wp2, 0);
        // in-place write of y = relu(pw1 + b1); 0 for halo pixels outside the image (the depthwise conv's zero padding)
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int rt = wave + 4 * r2;
            if (rt >= RT1) break;
            const int hp = rt * 32 + l31;
            const int hy = hp / HWd, hx = hp - hy * HWd;
            const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            const unsigned inmask = (unsigned)__ballot(hp < HP && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W);   // bits 0..31: lanes h = 0
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 32 + l31;
                if (col >= bf) continue;
                const float bias = a.b1[col];
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int row = (k & 3) + 8 * (k >> 2) + 4 * h;
                    float v = acc[r2][nt][k] + bias;
                    v = v > 0.0f ? v : 0.0f;
                    if (!((inmask >> row) & 1u)) v = 0.0f;
                    tile[(size_t)(rt * 32 + row) * CS + col] = v;
                }
            }
        }
    }
    __syncthreads();                                         // y complete, W1 buffers free
    stage_b(0);
    __syncthreads();

#ifdef YN_EXP_TIMING
    const long long T2 = __builtin_readcyclecounter();
#endif
    // ---- 3. pw2 on the interior; A fragment = depthwise 3x3 of y, evaluated from LDS --------------------------
    zero_acc();
    {
        const bool has1 = wave + 4 < RT2;
        int base0, base1;
        {
            int p = wave * 32 + l31; if (p >= OP) p = 0;
            const int oy = p / a.TW, ox = p - oy * a.TW;
            base0 = (oy * HWd + ox) * CS + 2 * h;
            p = (has1 ? wave + 4 : wave) * 32 + l31; if (p >= OP) p = 0;
            const int oy1 = p / a.TW, ox1 = p - oy1 * a.TW;
            base1 = (oy1 * HWd + ox1) * CS + 2 * h;
        }
        // depthwise fragment of one pixel for channels k + 2h, k + 2h + 1: 9 taps + bias from LDS
        auto dw_frag = [&](int base, int k, const float2 (&w)[10]) {
            float2 s = w[9];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float2 v = *reinterpret_cast<const float2*>(tile + base + ((tap / 3) * HWd + tap % 3) * CS + k);
                s.x += v.x * w[tap].x;
                s.y += v.y * w[tap].y;
            }
            if (a.dw_act == 1) { s.x = s.x > 0.0f ? s.x : 0.0f; s.y = s.y > 0.0f ? s.y : 0.0f; }
            return s;
        };
        auto frags = [&](int k, float2& f0, float2& f1) {
            float2 w[10];
#pragma unroll
            for (int tap = 0; tap < 10; ++tap) w[tap] = *reinterpret_cast<const float2*>(wd + tap * CS + k + 2 * h);
            f0 = dw_frag(base0, k, w);
            f1 = dw_frag(base1, k, w);
        };
        constexpr int VPM = (44 + 4 * NT - 1) / (4 * NT);    // VALU ops of the next fragments slotted after each MFMA
        for (int c = 0; c < nchunks; ++c) {
            const int buf = c & 1;
            if (c + 1 < nchunks) prefetch_b(a.Wp2, c + 1);
            const float* Bb = Bs + buf * KP * BS + l31 * 2 + h * BS;
            const int kc = c * 2 * KP;
            float2 a0, a1, bv[NT];
            frags(kc, a0, a1);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) bv[nt] = *reinterpret_cast<const float2*>(Bb + nt * 64);
#pragma unroll
            for (int q = 0; q < NQC; ++q) {
                float2 a0n = a0, a1n = a1, bvn[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bvn[nt] = bv[nt];
                __builtin_amdgcn_sched_barrier(0);
                if (q + 1 < NQC) {
                    frags(kc + 4 * (q + 1), a0n, a1n);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) bvn[nt] = *reinterpret_cast<const float2*>(Bb + 2 * (q + 1) * BS + nt * 64);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.x, bv[nt].x, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.x, bv[nt].x, acc[1][nt], 0, 0, 0);
                    acc[0][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0.y, bv[nt].y, acc[0][nt], 0, 0, 0);
                    acc[1][nt] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1.y, bv[nt].y, acc[1][nt], 0, 0, 0);
                }
                // schedule: every LDS read of step q+1 first, then one MFMA of step q followed by a slice of the depthwise FMAs
                if (q + 1 < NQC) __builtin_amdgcn_sched_group_barrier(0x100, 28 + NT, 0);
#pragma unroll
                for (int i = 0; i < 4 * NT; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (q + 1 < NQC) __builtin_amdgcn_sched_group_barrier(0x002, VPM, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                a0 = a0n; a1 = a1n;
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) bv[nt] = bvn[nt];
            }
            if (c + 1 < nchunks) { stage_b(buf ^ 1); __syncthreads(); }
        }
    }

#ifdef YN_EXP_TIMING
    const long long T3 = __builtin_readcyclecounter();
#endif
    // ---- 4. epilogue: out[2j] = x1[j], out[2j+1] = relu(pw2[j] + b2[j]) ------------------------------------
    // 4x4 transpose inside each quad of lanes (two xor-shuffles) => lane j of a quad holds row j x 4 consecutive channels:
    // 16-byte loads of x1 (unconditional, clamped pixel) and 16-byte stores of the interleaved result.
    {
        const int j = lane & 3;
        const size_t m_first = (size_t)y0 * a.W + x0;        // always a legal pixel
#pragma unroll
        for (int r2 = 0; r2 < 2; ++r2) {
            const int rt = wave + 4 * r2;
            if (rt >= RT2) break;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int col = nt * 32 + l31;
                const float bias = col < bf ? a.b2[col] : 0.0f;
                const int nq = nt * 32 + (l31 & ~3);          // first channel of the quad
                float4 pv[4];
                size_t mrow[4];
                bool okrow[4];
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int p = rt * 32 + 8 * g + 4 * h + j;
                    const int oy = p / a.TW, ox = p - oy * a.TW;
                    okrow[g] = p < OP && oy < th && ox < tw && nq < bf;
                    mrow[g] = okrow[g] ? (size_t)(y0 + oy) * a.W + (x0 + ox) : m_first;
                    pv[g] = *reinterpret_cast<const float4*>(xb + mrow[g] * C + (nq < bf ? nq : 0));
                }
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    float v0 = acc[r2][nt][4 * g + 0] + bias, v1 = acc[r2][nt][4 * g + 1] + bias;
                    float v2 = acc[r2][nt][4 * g + 2] + bias, v3 = acc[r2][nt][4 * g + 3] + bias;
                    v0 = v0 > 0.0f ? v0 : 0.0f; v1 = v1 > 0.0f ? v1 : 0.0f; v2 = v2 > 0.0f ? v2 : 0.0f; v3 = v3 > 0.0f ? v3 : 0.0f;
                    {
                        const float s0 = (j & 1) ? v0 : v1, s1 = (j & 1) ? v2 : v3;
                        const float q0 = __shfl_xor(s0, 1), q1 = __shfl_xor(s1, 1);
                        if (j & 1) { v0 = q0; v2 = q1; } else { v1 = q0; v3 = q1; }
                    }
                    {
                        const float s0 = (j & 2) ? v0 : v2, s1 = (j & 2) ? v1 : v3;
                        const float q0 = __shfl_xor(s0, 2), q1 = __shfl_xor(s1, 2);
                        if (j & 2) { v0 = q0; v1 = q1; } else { v2 = q0; v3 = q1; }
                    }
                    if (okrow[g]) {
                        float* o = ob + mrow[g] * C + 2 * nq;
                        *reinterpret_cast<float4*>(o) = make_float4(pv[g].x, v0, pv[g].y, v1);
                        if (nq + 2 < bf) *reinterpret_cast<float4*>(o + 4) = make_float4(pv[g].z, v2, pv[g].w, v3);
                    }
                }
            }
        }
    }
#ifdef YN_EXP_TIMING
    const long long T4 = __builtin_readcyclecounter();
    if (blockIdx.x == 5 && (t & 63) == 0)
        printf("unit bf=%d wave %d: load %lld  pw1 %lld  pw2+dw %lld  epilogue %lld [cycles]\n", bf, wave, T1 - T0, T2 - T1, T3 - T2, T4 - T3);
#endif
}

size_t shuffle_unit_lds(const UnitArgs& a, int NT, int KP)
{
    const int HP = (a.TH + 2) * (a.TW + 2);
    const int RT1 = (HP + 31) / 32;
    return ((size_t)RT1 * 32 * a.CS + 2 * KP * (32 * NT * 2) + 10 * a.CS + 8) * sizeof(float);
}

// returns false when the shape is not covered (the caller then runs the three separate kernels)
bool launch_shuffle_unit(const UnitArgs& a0, hipStream_t s)
{
    UnitArgs a = a0;
    const int NT = a.Npad / 32;
    if ((NT != 2 && NT != 4) || (a.bf & 1) || a.bf > a.Npad || a.bf / 2 > 256) return false;
    a.CS = a.bf + 2 + (((a.bf + 2) & 3) == 0 ? 2 : 0);       // pixel stride = odd number of 8-byte granules: conflict-free ds_read_b64
    // tile: at most 256 halo pixels and 256 interior pixels (two 32-row MFMA tiles per wave)
    int TH = 13, TW = 13;
    if (a.H < TH) TH = a.H;
    if (a.W < TW) TW = a.W;
    while ((TH + 2) * (TW + 2) > 256) --TH;
    a.TH = TH; a.TW = TW;
    a.tilesY = (a.H + TH - 1) / TH; a.tilesX = (a.W + TW - 1) / TW;
    const int KP = NT == 2 ? 8 : 16;                         // NT = 2: 74 KB of LDS => two blocks per CU
    const size_t lds = shuffle_unit_lds(a, NT, KP);
    if (lds > 160 * 1024) return false;
    static bool attr = false;
    if (!attr) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(shuffle_unit_kernel<2, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(shuffle_unit_kernel<4, 16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        attr = true;
    }
    const dim3 grid((unsigned)(a.B * a.tilesY * a.tilesX));
    set_last_kernel_name(NT == 2 ? "shuffle_unit_kernel<2,8>" : "shuffle_unit_kernel<4,16>");
    if (NT == 2) hipLaunchKernelGGL((shuffle_unit_kernel<2, 8>), grid, dim3(256), lds, s, a);
    else hipLaunchKernelGGL((shuffle_unit_kernel<4, 16>), grid, dim3(256), lds, s, a);
    return true;
}

}  // namespace ynk

// yn_h16.h — declarations of the fp16 training kernels (kernels_h16.hip) shared with the executor (yn_train_h16.inc).
#pragma once
#include "yn_internal.h"

namespace ynk {

typedef _Float16 h16;
// copies of every BatchNorm sum accumulator the fp16 step spreads its atomics over (same-address fp64 atomics serialise at the
// memory side: with 8 copies and 512 blocks the tail of a reduction was longer than its streaming phase).  Every workgroup of the
// BatchNorm apply / backward launches adds the copies up again in its prologue, so more copies are not free either: 32 / 16 / 8 copies =
// 7.42 / 7.29 / 7.32 ms per 608 x 608 bs-32 step (round 4, with the reductions at 256 workgroups and most sums taken in conv epilogues).
#ifndef YN_HACC_SLOTS
#define YN_HACC_SLOTS 16
#endif
constexpr int HACC_SLOTS = YN_HACC_SLOTS;
static_assert(HACC_SLOTS >= 1 && (HACC_SLOTS & (HACC_SLOTS - 1)) == 0, "YN_HACC_SLOTS must be a power of two: producers pick their copy with blockIdx & (HACC_SLOTS - 1)");

// Column sums of an hgemm output tile, taken in the kernel's epilogue while the tile sits in LDS (one launch and one pass over the
// tensor less per BatchNorm and direction):
//   forward (y == null):  acc[0][c] += sum out,  acc[1][c] += sum out^2      — the BatchNorm statistics of the conv output just computed;
//   input gradient (y != null): the tile is dz of the layer BELOW (the BN + activation whose output this conv consumed); with that layer's
//   pre-BN output y and saved statistics:  acc[0][c] += sum d,  acc[1][c] += sum d * xhat,  d = dz * act'(BN(y))  (hcol_reduce_kernel<2>'s sums).
// Physical column p of the tile <-> logical channel logical_of(p, C, half, gap); fp32 over the tile's 128 rows, double atomics per tile.
struct HColStat {
    double* acc;                                // [HACC_SLOTS][2][C]; null = no statistics
    int C, half, gap;
    const h16* y; int y_ld;
    const float* mean; const float* invstd; const float* gamma; const float* beta; int act;
};

// C[M][Np] (+)= A[M][Kp] (x taps) * Wp + bias — see hgemm_kernel
struct HGemmArgs {
    const h16* in; int in_ld, in_off;           // A rows: Kp physical channels at in + m*in_ld + in_off (16-byte aligned)
    int H, W, taps;                             // taps = 9: dense 3x3 stride 1 pad 1 over [B,H,W]; 1: pointwise
    const h16* Wp; const float* bias;           // packed [taps][Kp/8][Npad][8]; bias [Npad] or null
    h16* out; int out_ld, out_off;
    int M, Kp, Np, Npad;                        // Np = physical output channels written (multiple of 8), Npad = packed width (multiple of 32)
    int accumulate;                             // out += result (input-gradient accumulation)
    HColStat st;                                // epilogue statistics (st.acc == null: none); not with accumulate
};
void launch_hgemm(const HGemmArgs& a, hipStream_t s);

struct HWgradArgs {
    const h16* dy; int dy_ld;                   // [M][Np] dense
    const h16* x; int x_ld, x_off;              // [M][Kp] view
    int H, W, taps;
    int M, Np, Kp;                              // physical widths
    int N, Cin, half, gap;                      // logical Cout / Cin and the channel map of x (reference-layout scatter)
    float* dw;                                  // [N][Cin][taps] fp32 (written)
    float* partial; size_t partial_cap;
};
void launch_hwgrad(const HWgradArgs& a, hipStream_t s);

struct HDwArgs {
    const h16* in; int in_ld, in_off;
    const float* w; const float* bias;          // [9][Cp], [Cp] or null
    h16* out; int out_ld, out_off;
    int B, H, W, Cp, stride, accumulate;
    HColStat st;                                // stride 1 only: statistics of the output / BatchNorm-backward sums of the layer below (st.acc == null: none)
};
void launch_hdw(const HDwArgs& a, hipStream_t s);
void launch_hdw_dgrad_s2(const h16* dy, int dy_ld, const float* w, int B, int H, int W, int Cp, h16* dx, int dx_ld, int dx_off, int accumulate, hipStream_t s);
void launch_hdw_wgrad(const h16* dy, int dy_ld, const h16* x, int x_ld, int x_off, int B, int H, int W, int C, int Cp, int half, int gap, int stride,
                      float* dw /* [C][9], added to */, float* part /* scratch, part_cap floats */, size_t part_cap, hipStream_t s);

struct HRedArgs {
    const h16* y; int y_ld, y_off;              // the matrix (stats / column sum) or the pre-BN conv output (BN backward)
    int M, C, Cp, half, gap;                    // logical channels, physical width, channel map
    const h16* dz; int dz_ld, dz_off, dz_odd, dz_half, dz_gap;
    const float* mean; const float* invstd; const float* gamma; const float* beta; int act;
    double* acc; float* facc; size_t slot_stride;
    h16* even; int even_ld;                     // hbn_bwd_kernel, dz_odd only, optional: the EVEN logical channels of dz -> dense [M][even_ld] (pads zero)
    int lanes;                                  // filled by the launcher
};
void launch_hcol_reduce(const HRedArgs& a, int mode, hipStream_t s);      // mode 0 stats, 2 BN-backward sums, 3 column sum -> facc slots
void launch_hbn_bwd(const HRedArgs& a, h16* dy, float* dgamma, float* dbeta, hipStream_t s, bool sums_done = false);   // sums_done: acc already holds the sums (HColStat)

struct HBnApplyArgs {
    const h16* y; int y_ld; const double* acc; float eps;
    int M, C, Cp, half, gap, act;
    float* mean; float* invstd; const float* gamma; const float* beta;
    float* rmean; float* rvar; float momentum;
    h16* out; int out_ld, out_off;
    const h16* pass; int pass_ld, pass_off; int out_half, out_gap;      // shuffle mode
    int lanes;                                  // filled by the launcher
};
void launch_hbn_apply(const HBnApplyArgs& a, hipStream_t s);

void launch_hstem(const float* x_nchw, int B, int H, int W, const float* w, const float* bias, h16* y, hipStream_t s);
void launch_hstem_wgrad(const h16* dy, const float* x_nchw, int B, int H, int W, float* dw_slots, size_t slot_stride, hipStream_t s);
void launch_hmaxpool_idx(const h16* x, int B, int H, int W, int Cp, h16* y, uint8_t* idx /* window position of the maximum */, hipStream_t s);
void launch_hmaxpool_bwd(const h16* dy, const uint8_t* idx, int B, int H, int W, int Cp, h16* dx, hipStream_t s);
// the stem's BatchNorm + activation + max pool fused (no full-resolution normalised tensor, no full-resolution gradient): kernels_h16.hip
void launch_hstem_apply_pool(const HBnApplyArgs& a, int B, int H, int W, h16* out, uint8_t* idx, hipStream_t s);
void launch_hstem_bwd(const HRedArgs& a, const h16* pool_grad, const uint8_t* idx, int B, int H, int W, h16* dy, float* dgamma, float* dbeta, hipStream_t s);
void launch_hresample(const h16* a, const h16* b, h16* out, int B, int H, int W, int Cp, int mode, hipStream_t s);
void launch_hgather(const h16* src, int src_ld, int src_off, int src_cs, int src_half, int src_gap,
                    h16* dst, int dst_ld, int dst_off, int dst_cs, int dst_half, int dst_gap, long M, int n, int npad, hipStream_t s);
void launch_hpack_gemm(const float* w, int Cout, int Cin, int taps, int in_half, int in_gap, int Kp, int Npad, int backward, h16* out, hipStream_t s);
void launch_hpack_dw(const float* w, const float* bias, int C, int half, int gap, int Cp, int flip, float* out, float* bias_out, hipStream_t s);
void launch_hpack_stem(const float* w, float* out, hipStream_t s);
// one layer's packing job for hpack_all_kernel: kind 0 GEMM-shaped (fwd + bwd packs, bias), 1 depthwise (Kp = physical channels), 2 stem
struct HPackDesc {
    const float* w; const float* b;
    int kind, Cout, Cin, taps, half, gap, Kp, Npad, Kpb, Npadb;
    h16* wf; h16* wb; float* bias; float* dwf; float* dwb;
};
void launch_hpack_all(const HPackDesc* table_dev, int n, hipStream_t s);
void launch_hstage(const float* src, int C, h16* dst, int ld, int half, int gap, long M, hipStream_t s);
void launch_hunstage(const h16* src, int ld, int half, int gap, float* dst, int C, long M, hipStream_t s);
void launch_rows_to_f32(const void* src, int is_h16, int src_ld, float* dst, int n, long M, hipStream_t s);
void launch_hgrad_finish(float* g, const float* slots, long n, size_t stride, float* state, hipStream_t s);
// settle the pending fp16 step's loss-scale decision: from yn_sgd_step's bucket-wide non-finite flag, or (null) from the local one
void launch_hscale_update(float* state, const int* global_flag, hipStream_t s);
// the loss on fp16 head tensors (kernels_train.hip): gradients are multiplied by the loss scale state[0] read on the device
void launch_loss_h16(const h16* const head[3], h16* const ghead[3], const float* target, const GridInfo& g, int B, float* partial, float* losses,
                     const float* scale_state, hipStream_t s);

}  // namespace ynk

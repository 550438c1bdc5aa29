// yn_internal.h — shared declarations between the C-ABI layer (yn_api.hip) and the gfx950 kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ynk {

// Every launch of the library goes through this form of hipLaunchKernelGGL: with YN_LOG_LDS=1 in the environment each distinct (kernel, dynamic
// LDS bytes, threads) is written to stderr once ("yn_lds <kernel> <bytes> <threads>").  The rocprofv3 kernel trace reports a kernel's STATIC LDS
// only; tools/concurrency.py joins these lines (profiles/r05_dynamic_lds.txt) to get the real per-workgroup footprint.
extern bool g_log_lds;
void note_launch_lds(const char* kernel, size_t dyn_lds, unsigned threads);
}  // namespace ynk
#undef hipLaunchKernelGGL
#define hipLaunchKernelGGL(kernelName, numBlocks, numThreads, memPerBlock, streamId, ...)                                            \
    do {                                                                                                                              \
        if (ynk::g_log_lds) ynk::note_launch_lds(#kernelName, (size_t)(memPerBlock), (unsigned)(dim3(numThreads).x * dim3(numThreads).y)); \
        hipLaunchKernelGGLInternal((kernelName), numBlocks, numThreads, memPerBlock, streamId, __VA_ARGS__);                         \
    } while (0)
namespace ynk {

// ---- GEMM-shaped convolutions (pointwise 1x1 and dense 3x3) on the f32 MFMA ----------------------
// A operand = activations, NHWC: row m (= pixel) has K contiguous floats at in + m*in_ld + in_off.
// B operand = folded weights packed k-pair interleaved: Wp[(k/2)][n][k&1], n in [0,Npad), zero padded.
struct GemmArgs {
    const float* in;   int in_ld;   int in_off;
    const float* in2;  int resample;            // conv3x3 only: 0 none, 1 += up2(in2), 2 += down2(in2)
    int H, W;                                   // conv3x3 only: spatial extent (M = B*H*W)
    const float* Wp;   const float* bias;       // [Kp/2][Npad][2], [Npad]
    float* out;        int out_ld;  int out_off;
    const float* pass; int pass_ld; int pass_off; // shuffle mode: out[m][2n] = pass[m][n], out[m][2n+1] = res
    int M, K, N, Npad, act;
    int cfg;                                    // pointwise tile configuration index, -1 = heuristic
    const float* dw_w; const float* dw_b;       // fused depthwise prologue (dwpw kernel): [9][K], [K]
    int dw_act;                                 // activation between the depthwise and the pointwise conv
    int dw_stride;                              // dwpw_tile_kernel: stride of the depthwise conv (H, W = its INPUT extent, M = output pixels)
    const void* Wsh; const void* Wsl;           // split-f16 packs of the same weights (hi, lo * 2^11): [taps][ceil(Cin/8)][Npad][8] halves, or null.
                                                // INVARIANT the kernels rely on (their prefetches carry no masks): every value of every octet
                                                // < ceil(Cin/8) and every column < Npad is FINITE (fold_pack_kernel checks what it packs and zero-fills the
                                                // padding) - the A operand's zero K tail times a clamped / padded B octet must be an exact zero; and
                                                // accumulator rows >= M hold duplicates of row M - 1: an epilogue must never reduce over rows
    unsigned* ovf;                              // split-f16 range guard (yn_device.h, range_report): set to 1 when an activation >= 65504 was split; or null
    int in_slack;                               // bytes that may be READ past the last element of `in` (arena tensors: 16; caller's tensors: 0) - pw_pipe_kernel's
                                                // 16-byte DMA pieces run 8 bytes past a row whose K is not a multiple of 4
};

struct DwArgs {
    const float* in;  int in_ld;  int in_off;
    const float* w;   const float* bias;        // [9][C], [C]
    float* out;       int out_ld; int out_off;
    int B, H, W, C, stride, act;                // H,W = input extent
};

// One kernel per stride-1 ShuffleV2 unit, cut at the unit's depthwise conv instead of at its input (kernels_conv.hip,
// unit_chain_kernel): depthwise 3x3 -> pw2 -> concat+shuffle -> (the NEXT unit's pw1).
struct ChainArgs {
    const float* t1; int t1_ld, t1_off;         // depthwise input = this unit's pw1 output, [M][bf]
    const float* x1; int x1_ld, x1_off;         // pass-through half of the unit's input, [M][bf]
    const float* wdw; const float* bdw; int dw_act;     // depthwise [9][bf], [bf] (BN folded)
    const float* Wp2; const float* b2; int act2;        // pw2: packed [bf/2][Npad][2], bias [Npad]
    const float* Wp1n; const float* b1n; int act1n;     // next unit's pw1 (null: last unit of the stage)
    const void *Ws2h, *Ws2l, *Ws1h, *Ws1l;              // split-f16 packs of pw2 / the next pw1 (null: the f32-MFMA chain kernel)
    float* out; int out_ld;                     // next != null: first half of the shuffled output, [M][bf]; else the whole output [M][2*bf]
    float* t1n;                                 // next unit's depthwise input [M][bf]
    int B, H, W, bf, Npad, M;
    unsigned* ovf;                              // split-f16 range guard flag or null
    int pipe_mode;                              // unit_pipe_kernel: 0 by the size rule, 1 never, 2 also for few tiles (yn_chain_pipe)
};
// The stride-1 units of a stage (all but the last) as ONE persistent launch (kernels_stage.hip, stage_pipe_kernel): work items = (unit, tile)
// handed out by ticket, tile-level ready flags between the units.
struct StageUnit {
    const float* t1;                            // depthwise input = this unit's pw1 output, [M][bf]
    const float* x1; int x1_ld; int pad_;       // pass-through half of the unit's input (offset 0)
    const float* wdw; const float* bdw;         // depthwise [9][bf], [bf]
    const float* b2; const float* b1n;          // biases of pw2 and of the next unit's pw1
    const void *Ws2h, *Ws2l, *Ws1h, *Ws1l;      // split-f16 packs of pw2 / the next unit's pw1
    float* out;                                 // first half of the shuffled output [M][bf] (the next unit's pass-through half)
    float* t1n;                                 // the next unit's depthwise input [M][bf]
};
constexpr int YN_STAGE_MAX = 7;
struct StageArgs {
    StageUnit u[YN_STAGE_MAX];
    int nunits, M, H, W, tiles;                 // tiles: set by launch_stage_pipe (the form's tile height)
    float inv_w, inv_h;
    unsigned* ovf;                              // split-f16 range guard flag or null
    unsigned* sync;                             // stage_sync_words() words, zero between launches (the kernel leaves them zero)
};
// layout of the sync words: eight queue heads (one per 256 bytes), the exit count, the timeout mark (set when a bounded wait expired: the results of
// that launch are not to be trusted - yn_range_status reports and clears it), then one ready flag per (unit, tile)
constexpr int STAGE_HEAD_STRIDE = 64;
constexpr int STAGE_EXIT = 8 * STAGE_HEAD_STRIDE;
constexpr int STAGE_TIMEOUT = STAGE_EXIT + 1;
constexpr int STAGE_FLAGS = 1024;
size_t stage_sync_words(int tiles, int nunits);
bool launch_stage_pipe(StageArgs a, int bf, int pub_early, int min_tiles, size_t sync_bytes, hipStream_t s, bool dry = false);   // false = no form for this shape / too few tiles, nothing launched
bool launch_unit_chain(const ChainArgs& a, hipStream_t s);
bool launch_unit_pipe(const ChainArgs& a, hipStream_t s, bool dry = false);   // kernels_pipe.hip: the persistent tile walk; false = not applicable, nothing launched

// The main branch of a stride-2 ShuffleV2 unit as ONE kernel (kernels_chain.hip, down_unit_kernel): pw1 -> depthwise 3x3 stride 2 ->
// pw2 -> concat + shuffle with the other branch's output.
struct DownArgs {
    const float* x; int cin;                    // unit input [B][H][W][cin] (dense)
    const void *W1h, *W1l; const float* b1; int act1, Npad1;       // pw1: split packs [ceil(cin/8)][Npad1][8], bias, activation
    const float* wdw; const float* bdw; int dw_act;                // depthwise [9][bf], [bf]
    const void *W2h, *W2l; const float* b2; int act2, Npad2;       // pw2: split packs [ceil(bf/8)][Npad2][8]
    const float* pass;                          // branch-1 output [B][Ho][Wo][bf], or null: branch 1 is computed here too (the members below)
    const float* wdw1; const float* bdw1; int dw1_act;             // branch 1 depthwise (stride 2, on x): [9][cin], [cin]
    const void *W3h, *W3l; const float* b3; int act3, Npad3;       // branch 1 pointwise cin -> bf
    float* out;                                 // [B][Ho][Wo][2*bf]: out[2n] = pass[n], out[2n+1] = pw2[n]
    int B, H, W, bf;
    unsigned* ovf;                              // split-f16 range guard flag or null
};
// A stride-2 ShuffleV2 unit AFTER its first pointwise conv (kernels_chain.hip, down2_kernel): both depthwise convs (stride 2), both
// pointwise convs behind them and the concat + shuffle as one kernel; y1 = act(pw1(x)) comes from a gemm_split_kernel launch.
struct Down2Args {
    const float* x; int cin;                    // unit input [B][H][W][cin] (dense)
    const float* y1;                            // act(pw1(x)) [B][H][W][bf] (dense)
    const float* wdw; const float* bdw; int dw_act;                // branch 2 depthwise [9][bf], [bf]
    const void *W2h, *W2l; const float* b2; int act2;              // pw2: split packs [ceil(bf/8)][Npad][8], bias
    const float* wdw1; const float* bdw1; int dw1_act;             // branch 1 depthwise (on x): [9][cin], [cin]
    const void *W3h, *W3l; const float* b3; int act3;              // branch 1 pointwise cin -> bf: split packs [ceil(cin/8)][Npad][8]
    float* out;                                 // [B][Ho][Wo][2*bf]: out[2n] = branch 1, out[2n+1] = branch 2
    const void *W1nh, *W1nl; const float* b1n; int act1n;          // the NEXT unit's first pointwise conv (bf -> bf, on channels [bf, 2bf) of out), or null
    float* t1n;                                 // its output [B][Ho][Wo][bf] (the next unit's depthwise input)
    int B, H, W, bf, Npad;
    unsigned* ovf;                              // split-f16 range guard flag or null
};
bool down2_covers(const Down2Args& a);
void launch_down2(const Down2Args& a, hipStream_t s);
// depthwise 3x3 (stride 1) + pointwise conv of a detection head as one kernel (kernels_chain.hip, dwpw_group_kernel): C = Cout = 96
struct DwPwArgs {
    const float* in;                            // [B][H][W][C] dense
    const float* wdw; const float* bdw; int dw_act;                // depthwise [9][C], [C]
    const void *Wh, *Wl; const float* bias; int act, Npad;         // pointwise: split packs [C/8][Npad][8]
    float* out;                                 // [B][H][W][C] dense
    int B, H, W, C;
    unsigned* ovf;                              // split-f16 range guard flag or null
};
bool dwpw_group_ok(const DwPwArgs* a, int n);
void launch_dwpw_group(const DwPwArgs* a, int n, hipStream_t s);
bool down_unit_covers(const DownArgs& a);
void launch_down_unit(const DownArgs& a, hipStream_t s);
bool unit_chain_covers(const ChainArgs& a);     // same selection, nothing launched

const char* last_kernel_name();            // symbol of the most recent launch_* on this thread
void set_last_kernel_name(const char* n);
int  pw_config_count();                    // f32-MFMA family (tiled, then register-direct) followed by the split-f16 family
int  pw_f32_config_count();
void launch_pw(const GemmArgs& a, hipStream_t s);
bool launch_pw_pipe(const GemmArgs& a, hipStream_t s);      // kernels_pipe.hip: the persistent form (the last pointwise configuration index); false = not applicable
void launch_conv3x3(const GemmArgs& a, hipStream_t s);
void launch_dw(const DwArgs& a, hipStream_t s);
void launch_stem(const float* x_nchw, int B, int H, int W, const float* w /*[27][Cout]*/, const float* bias,
                 int Cout, int act, float* y, hipStream_t s);
void launch_maxpool(const float* x, int B, int H, int W, int C, float* y, hipStream_t s);
// stem conv + max pool fused: x NCHW [B,3,H,W] -> y NHWC [B,H/4,W/4,Cout]
void launch_stem_pool(const float* x_nchw, int B, int H, int W, const float* w, const float* bias, int Cout, int act, float* y, hipStream_t s);
void launch_nchw_to_nhwc(const float* x, int B, int C, int H, int W, float* y, hipStream_t s);
void launch_nhwc_to_nchw(const float* x, int B, int C, int H, int W, float* y, hipStream_t s);

// ---- weight preparation (BN folding + packing), device side --------------------------------------
struct FoldArgs {
    const float* w; const float* b;             // raw conv weight [Cout][per_out], bias or null
    const float* gamma; const float* beta; const float* mean; const float* var;  // null => no BN
    float eps;
    int Cout, Cin, kk;                          // kk = k*k taps; per_out = Cin_per_group*kk
    int kind;                                   // 0 pw/dense (GEMM pack), 1 depthwise, 2 stem
    int Kp, Npad;                               // GEMM pack geometry
    float* w_ref; float* b_ref;                 // folded, reference layout (for yn_get_folded) or null
    float* w_packed; float* b_packed;
    void* ws_hi; void* ws_lo;                   // GEMM kinds: split-f16 packs [kk][ceil(Cin/8)][Npad][8] (zero-initialised by the caller), or null
    unsigned* w_ovf;                            // set to 1 when a folded GEMM weight does not fit the split (|w| >= 65504 or non-finite), or null
};
void launch_fold_pack(const FoldArgs& a, hipStream_t s);

// ---- score head / NMS ---------------------------------------------------------------------------
struct GridInfo {
    int S, C, A, N;
    int hw[3], w[3], off[3];                    // cells per scale, width per scale, candidate offset
    int head_ld;                                // row stride (floats) of the raw head tensors: A(5+C), or padded to a multiple of 4
    float anchors[18];
};

void launch_score_full(const float* const heads[3], const GridInfo& g, int B, float* all_bbox, float* all_class, hipStream_t s);
void launch_decode_boxes(const float* txtytwth, const GridInfo& g, int B, float* xyxy, hipStream_t s);
// candidates from raw heads: per candidate best score / class / box
void launch_decode_cand(const float* const heads[3], const GridInfo& g, int B, float conf_thresh,
                        float* boxes, float* scores, int32_t* cls, hipStream_t s);
// the last pointwise conv of head `scale` + the decode of its candidates in one kernel (split-f16 packs, A(5+C) <= 256 columns)
bool head_decode_supported(const GemmArgs& a, const GridInfo& g);
void launch_head_decode(const GemmArgs& a, const GridInfo& g, int scale, float conf_thresh,
                        float* boxes, float* scores, int32_t* cls, hipStream_t s);
// candidates from (all_local, all_conf): argmax + threshold (models/yolo_nano.py:253-261)
void launch_argmax_cand(const float* all_local, const float* all_conf, int B, int N, int C, float conf_thresh,
                        float* boxes, float* scores, int32_t* cls, hipStream_t s);
struct NmsWork {                                // per-handle scratch, sized for B*N candidates / B*C segments
    int32_t* seg_count;                         // [B][C]
    int32_t* seg_off;                           // [B][C]
    int32_t* tile_off;                          // [B][C+1]  first 64x64 tile of each segment, [C] = total
    int32_t* bucket;                            // [B][N]    candidate ids grouped by class, then sorted by score
    int32_t* keep;                              // [B][N]    flags
    float*   sbox;                              // [B][N][4] boxes in sorted order
    void*    matrix;                            // [B][matrix_stride] uint64 suppression bit-matrix tiles
    size_t   matrix_stride;                     // words per image
    int32_t* large_list;                        // [B][large_cap+1]  count, then class ids of segments with n > 1024
    int32_t* bucket2; float* sbox2;             // [B][N], [B][N][4]: the candidates that survive the first-chunk prefilter, per segment at seg_off
    int32_t* seg_count2; int32_t* tile_off2;    // [B][C], [B][C+1] of that list
    int32_t* seg_order;                         // [B][C]  the image's class ids by segment size, largest first (bucket_kernel)
    int32_t* seg_sparse;                        // [B][C]  1 = the segment's suppression words come from nms_sweep_kernel (matrix_kernel only zeroes its tiles)
    int32_t* work_off;                          // [B][C+1] tile_off2 without those segments: matrix_kernel's work list
    void*    pre_sync;                          // [nms_pre_sync_words(B, N)] uint64: nms_prefilter_kernel's sliced large segments - ticket counter (zero between launches) + survivor words
    int32_t* ctr;                               // [B][2]  bucket_sort_kernel's position cursor / finished-workgroup count (zero between launches)
    int      prefilter;                         // 0 off, 1 for batches of >= 4 images, 2 always
    int      sweep;                             // 1: spread-out large segments on nms_sweep_kernel (behind the prefilter), 0: every segment on matrix_kernel
    int      large_cap;
    const unsigned* ovf;                        // yn_infer: the split-f16 range flag of the network kernels that produced the candidates, or null.
                                                // Set => compact_kernel reports count[b] = -1 - kept (the results are invalid: yn_range_status)
    unsigned* ovf_host;                         // with ovf: one word of pinned host memory (device view) that compact_kernel sets to 1 beside the negative counts, or null
};
size_t nms_matrix_words_per_image(int N, int C);
size_t nms_pre_sync_words(int B, int N);
int nms_max_segment();                      // largest per-class segment resolve_segment() can hold (its removed-mask lives in LDS)
// optional per-kernel hook of launch_nms_pipeline: called with the kernel's name right before each launch (profiling brackets)
struct NmsHook { void (*fn)(void* ctx, const char* kernel); void* ctx; };
void launch_nms_pipeline(const float* boxes, const float* scores, const int32_t* cls, int B, int N, int C,
                         float nms_thresh, int diou, const NmsWork& wk,
                         float* out_boxes, float* out_scores, int32_t* out_cls, int32_t* out_index, int32_t* count,
                         hipStream_t s, const NmsHook* hook = nullptr);
void launch_pack(const float* boxes, const float* scores, const int32_t* cls, const int32_t* count, int B, int N, float* rec, int32_t* offsets, hipStream_t s);
void launch_nms_single(const float* dets, const float* scores, int n, float thresh, int diou,
                       int32_t* ids_scratch, float* sbox_scratch, void* matrix_scratch, int32_t* keep, int32_t* count, hipStream_t s);

// ---- training loss (kernels_train.hip) --------------------------------------------------------------
int  loss_num_blocks(const GridInfo& g, int B);
void launch_loss(const float* conf, const float* cls, const float* t, const float* const head[3], float* const ghead[3],
                 const float* target, const GridInfo& g, int B, float* partial, float* losses,
                 float* g_conf, float* g_cls, float* g_t, hipStream_t s);

void launch_make_targets(const double* labels, const int32_t* offsets, int B, const double* anchors18, const GridInfo& g, float* target, hipStream_t s);

// ---- train-mode kernels (kernels_bwd.hip) --------------------------------------------------------------
// Same-address atomics serialise (~50 ns each on this part), so every atomically-accumulated result has several copies
// ("slots", chosen by block index) that the consumer sums: BatchNorm sums ACC_SLOTS x [2][C] doubles, weight gradients
// GRAD_SLOTS copies of the flat gradient buffer combined once per step by launch_grad_combine.
constexpr int ACC_SLOTS = 8;
constexpr int GRAD_SLOTS = 8;
static_assert((ACC_SLOTS & (ACC_SLOTS - 1)) == 0 && (GRAD_SLOTS & (GRAD_SLOTS - 1)) == 0, "slot copies are chosen with blockIdx & (SLOTS - 1)");
struct BnApplyArgs {
    const float* y; const double* acc; float eps;           // acc[ACC_SLOTS][2][C]: sum y, sum y*y (launch_bn_stats)
    float* mean; float* invstd;                              // saved for the backward pass
    const float* gamma; const float* beta;
    float* out; int out_ld, out_off, out_cs;
    const float* pass; int pass_ld, pass_off, pass_dst_off;
    float* rmean; float* rvar; float momentum;
    int M, C, act, lanesC;
};
struct BnBwdArgs {
    const float* dz; int dz_ld, dz_off, dz_cs;
    const float* y; const float* mean; const float* invstd; const float* gamma; const float* beta;
    double* acc;                                             // acc[ACC_SLOTS][2][C], zeroed: sum dyh, sum dyh*xhat
    float* dgamma; float* dbeta;
    float* dy;
    int M, C, act, lanesC;
};
struct WgradArgs {
    const float* dy; int dy_ld;
    const float* x; int x_ld, x_off;
    int H, W, Cin, dense;
    float* dw;                                               // the gradient itself (written, not accumulated)
    float* partial; size_t partial_cap;                      // scratch for the per-slice copies of dW (floats)
    int M, N, K;
};
void launch_bn_stats(const float* y, int M, int C, double* acc, hipStream_t s);
void launch_col_sum_accumulate(const float* x, int ld, int off, int M, int C, float* out, size_t slot_stride, hipStream_t s);
void launch_grad_combine(float* g, const float* slots, long n, size_t stride, hipStream_t s);
void launch_bn_apply(const BnApplyArgs& a, hipStream_t s);
void launch_bn_bwd(const BnBwdArgs& a, hipStream_t s);
void launch_wgrad(const WgradArgs& a, hipStream_t s);
void launch_dw_wgrad(const float* dy, const float* x, int x_ld, int x_off, int B, int H, int W, int C, int stride, float* dw /* [C][9], added to */,
                     float* part /* scratch */, size_t part_cap /* floats */, hipStream_t s);
void launch_dw_dgrad_s2(const float* dy, const float* w, int B, int H, int W, int C, float* dx, int dx_ld, int dx_off, int accumulate, hipStream_t s);
void launch_stem_wgrad(const float* dy, const float* x, int B, int H, int W, int Cout, float* dw, float* partial, size_t partial_cap, hipStream_t s);
void launch_maxpool_idx(const float* x, int B, int H, int W, int C, float* y, int32_t* idx, hipStream_t s);
void launch_maxpool_bwd(const float* dy, const int32_t* idx, int B, int H, int W, int C, float* dx, hipStream_t s);
void launch_resample(const float* a, const float* b, float* out, int B, int H, int W, int C, int mode, hipStream_t s);
void launch_strided_copy(const float* src, int src_ld, int src_off, int src_cs, float* dst, int dst_ld, int dst_off, int dst_cs,
                         long M, int n, int accumulate, hipStream_t s);
void launch_pack_bwd(const float* w, int Cout, int Cin, int kind, int Npad, float* out, hipStream_t s);

void launch_preprocess(const unsigned char* img, int h0, int w0, int rw, int rh, int left, int top, int side,
                       const float* mean, const float* stdv, float* out, hipStream_t s);
void launch_preprocess_batch(int n, const unsigned char* const* imgs, const int* geom, int side, const float* mean, const float* stdv,
                             float* out, hipStream_t s);
void launch_ema(float* v, const float* m, long n, float d, float one_minus_d, hipStream_t s);
// flag: int[2] on the device or null — [0] set when g holds a NaN/Inf (the update is then skipped), [1] counts skipped steps
void launch_sgd(float* p, const float* g, float* buf, long n, float lr, float momentum, float wd, float grad_scale, int first, int* flag, hipStream_t s);

// XCD-aware tile order.  Workgroup b runs on XCD b % 8 and every XCD has its own L2, so with the identity mapping the
// three input rows of a 3x3 window are fetched by three different L2s (PMC: FETCH_SIZE = 3.3x the input for the stage-3
// depthwise convs).  With the grid rounded up to a multiple of 8 (xcd_grid) this bijection hands every XCD one
// CONTIGUOUS eighth of the tile range, so only the rows at the seven seams are fetched twice.
__device__ __forceinline__ unsigned xcd_block(unsigned b, unsigned nb) { return (b & 7u) * (nb >> 3) + (b >> 3); }
inline unsigned xcd_grid(unsigned blocks) { return (blocks + 7u) & ~7u; }

// ---- grouped launches: up to three problems of one kernel type (the three detection heads' layer k, the three FPN laterals) as ONE
//      launch.  Workgroup ids [first[p], first[p+1]) belong to problem p; every range starts at a multiple of 8, so a problem's local
//      ids keep the id % 8 -> XCD relation the tile decode relies on.  The small problems ride along with the large one instead of
//      paying a launch each (a launch of this network's small layers costs 7-15 us whatever it computes).
#define YN_GROUP_MAX 3
template <typename Args> struct Group { Args a[YN_GROUP_MAX]; unsigned first[YN_GROUP_MAX + 1]; };
__device__ __forceinline__ int group_problem(const unsigned (&first)[YN_GROUP_MAX + 1], unsigned bid, unsigned& local, unsigned& nb)
{
    const int p = (bid >= first[1] ? 1 : 0) + (bid >= first[2] ? 1 : 0);
    local = bid - first[p];
    nb = first[p + 1] - first[p];
    return p;
}
void launch_dw_group(const DwArgs* a, int n, hipStream_t s);                    // stride 1, 4-channel vectors (checked by the caller through dw_group_ok)
bool dw_group_ok(const DwArgs* a, int n);
bool launch_pw_group(const GemmArgs* a, int n, int cfg, hipStream_t s);          // split-f16 family only; cfg = pointwise configuration index or -1
void launch_head_decode_group(const GemmArgs* a, int n, const GridInfo& g, float conf_thresh, float* boxes, float* scores, int32_t* cls, hipStream_t s);
// the tail of a detection head as one kernel: depthwise 3x3 + pointwise conv (layers .2 + .3) + last conv (.4) + candidate decode
struct HeadTailArgs {
    const float* in;                                    // [B][H][W][96] dense (the output of layers .0 + .1)
    const float* wdw; const float* bdw; int dw_act;     // depthwise [9][96], [96]
    const void *Wh, *Wl; const float* bias; int act;    // pointwise 96 -> 96: split packs [12][96][8]
    const void *Wfh, *Wfl; const float* fbias; int Npad;   // last conv 96 -> A(5+C): split packs [12][Npad][8], bias [Npad]
    int B, H, W;
    unsigned* ovf;                                      // split-f16 range guard flag or null
};
bool head_tail_ok(const HeadTailArgs* q, int n, const GridInfo& g);
void launch_head_tail_group(const HeadTailArgs* q, int n, const GridInfo& g, float conf_thresh, float* boxes, float* scores, int32_t* cls, hipStream_t s);

// hipFuncSetAttribute acts on the CURRENT device: every call site keeps a bit mask of the devices it has configured
inline bool attr_pending(unsigned long long& mask)
{
    int d = 0;
    (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    if (mask & bit) return false;
    mask |= bit;
    return true;
}

}  // namespace ynk

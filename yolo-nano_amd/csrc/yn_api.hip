// yn_api.hip — the C ABI of libyolonano_hip.so (see include/yolonano_hip.h) and the network executor.
//
// The handle owns: the raw parameters (reference state-dict keys), the folded + packed weights, one
// activation arena in HBM, NMS scratch, and optional hipGraphs of the fixed-shape pipelines.
// Network wiring restates models/yolo_nano.py:282-301 and backbone/shufflenetv2.py:69-78,157-167.
#include <hip/hip_runtime.h>

#include <dlfcn.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <vector>

#include "../../include/yolonano_hip.h"
#include "yn_internal.h"
#include "yn_h16.h"
#include <stdlib.h>

using namespace ynk;

namespace {

thread_local std::string g_create_error;
std::mutex g_tune_mutex;
std::map<std::vector<int>, int> g_pw_tuned;        // (device, M, K, N, strides...) -> fastest pointwise tile configuration

struct Param {
    void* dev = nullptr;
    size_t bytes = 0;
    size_t numel = 0;
    std::vector<int64_t> shape;
};

enum { K_PW = 0, K_DW = 1, K_DENSE3 = 2, K_STEM = 3 };

struct Layer {
    std::string name, conv, bn;
    int kind = 0, cin = 0, cout = 0, stride = 1, has_bias = 0, act = 0;
    float* w_packed = nullptr;
    float* b_packed = nullptr;
    float* w_ref = nullptr;
    float* b_ref = nullptr;
    void* ws_hi = nullptr;                 // split-f16 packs of the folded weights (dense 3x3: conv3x3_split_kernel)
    void* ws_lo = nullptr;
    size_t ws_bytes = 0;
    size_t w_numel = 0;
    int Kp = 0, Npad = 0;
};

struct TrainPack { float* wp = nullptr; float* bias = nullptr; float* wp_bwd = nullptr; int Kb = 0, Npad_b = 0; };   // per-layer training weight packs

// per-layer weight packs of the fp16 training step (yn_train_h16.inc): f16 GEMM packs (forward / input gradient), fp32 depthwise taps
struct HPack { ynk::h16* wf = nullptr; ynk::h16* wb = nullptr; float* bias = nullptr; float* dwf = nullptr; float* dwb = nullptr; int Kp = 0, Npad = 0, Kpb = 0, Npadb = 0; };

struct ProfRec {
    std::string name, kernel;
    hipEvent_t e0, e1;
    double flops, bytes;
};

struct TrainGraph { std::vector<uintptr_t> key; hipGraphExec_t exec; int direct_runs; };      // the fp16 training step's body as a graph (yn_train_h16.inc)
struct GraphEntry {
    std::vector<uintptr_t> key;
    hipGraphExec_t exec = nullptr;
};

}  // namespace

struct yn_handle {
    yn_config cfg;
    hipStream_t stream = nullptr;
    hipStream_t cur = nullptr;            // stream the launch helpers currently target (main or a side stream)
    hipStream_t side[2] = {nullptr, nullptr};
    std::vector<hipEvent_t> fj_events;    // fork/join events (rotating pool)
    size_t fj_next = 0;
    bool multi_stream = true;
    std::string err;
    std::map<std::string, Param> params;
    std::vector<Layer> layers;
    std::map<std::string, int> by_name;
    std::map<std::string, int> by_conv;
    bool folded = false;
    int stage_ch[3] = {0, 0, 0};
    int head_ch = 0;
    GridInfo grid;
    // arena
    char* arena = nullptr;
    size_t arena_bytes = 0, arena_used = 0;
    int arena_S = 0, arena_B = 0;
    // NMS scratch + candidate buffers (sized max_batch * N)
    float* cand_boxes = nullptr; float* cand_scores = nullptr; int32_t* cand_cls = nullptr;
    NmsWork nms{};
    size_t nms_cap = 0;           // elements B*N currently allocated
    size_t nms_seg_cap = 0;       // B*(C+1)
    size_t nms_m_cap = 0;         // uint64 words of suppression matrix
    size_t nms_ps_cap = 0;        // uint64 words of the prefilter's slice sync
    float* heads_int[3] = {nullptr, nullptr, nullptr};
    size_t heads_cap = 0;
    float* fwd_only[3] = {nullptr, nullptr, nullptr};     // yn_train_forward: the training executors stop after the forward pass and copy the raw heads here
    float* tap_out[3] = {nullptr, nullptr, nullptr};      // yn_forward_taps: where run_network copies c3 / c4 / c5
    float* loss_partial = nullptr;
    size_t loss_partial_cap = 0;
    // training (yn_train.inc): caller-owned flat buffers + per-layer packs + workspace
    float *tP = nullptr, *tG = nullptr, *tM = nullptr;
    int64_t tN = 0, tN_expected = 0;
    long train_steps = 0;
    std::map<std::string, size_t> toff;
    std::vector<TrainPack> tpacks;
    float* zeros = nullptr;
    float* train_losses = nullptr;        // device float[4]: the step's body writes its losses here (a fixed address for the graph), copied to the caller's buffer afterwards
    std::vector<TrainGraph> train_graphs; int train_graph_misses = 0; bool train_graph = false; int64_t train_graph_replays = 0;   // yn_train_graph
    int head_fork = -1, head_fork_now = 0, fork_trials = 0;      // -1: undecided (steps 3-6 time the step with and without the forks), else the choice
    hipEvent_t fork_ev[8] = {};           // timing events of the trials
    hipStream_t train_fork[2] = {nullptr, nullptr};      // the head towers of levels 3 / 4 (forward and backward) beside the main stream
    hipStream_t train_side = nullptr;     // the fp16 step's weight-gradient stream (lowest priority: the main stream is the critical path)
    int* skip_flag = nullptr;             // device int[2]: [0] this step's gradient is non-finite, [1] number of skipped updates
    int train_dtype = 0;                  // 0 fp32, 1 fp16 storage + f16 MFMA (yn_train_precision)
    std::vector<HPack> hpacks;
    std::vector<ynk::HPackDesc> hpack_jobs;   // recorded while the first fp16 step packs layer by layer; later steps pack everything in one launch
    ynk::HPackDesc* hpack_table = nullptr;     // device copy
    int hpack_table_n = 0;
    float* scale_state = nullptr;         // device float[8]: loss scale, its inverse, clean-step counter, local overflow flag, step-pending mark (kernels_h16.hip)
    float loss_scale_init = 0.0f, loss_scale_clean = 0.0f;   // yn_train_set_loss_scale before the first fp16 step (checkpoint resume)
    std::vector<hipEvent_t> train_events;
    char* train_arena = nullptr;
    size_t train_arena_bytes = 0;
    // graphs / profiling
    bool use_graph = false;
    long net_passes = 0;                   // run_network calls so far (dbg_skip)
    bool dwpw_fuse = true;                 // YN_DWPW_FUSE=0: the heads' depthwise + pointwise pairs as two grouped launches instead of one kernel
    bool tail_fuse = true;                 // YN_TAIL_FUSE=0: layers .2+.3 and .4+decode of the heads as two grouped kernels instead of one (head_tail_group_kernel)
    bool down_fuse = true;                 // yn_down_fuse / YN_DOWN_FUSE=0: the main branch of a stride-2 unit as one kernel (down_unit_kernel)
    bool group_launch = true;              // yn_group_launch / YN_GROUP=0: the three heads' layers (and the laterals) as grouped launches
    bool fuse_decode = true;               // yn_fuse_decode / YN_FUSE_DECODE=0: yn_infer's last head conv + candidate decode as one kernel
    int fuse_decode_mode = 1;              // 1 = when the stride-8 head has >= 8192 pixels, 2 = always
    bool exact_f32 = false;                // yn_exact_f32 / YN_EXACT_F32=1: GEMM-shaped convs on the f32 MFMA only (no split-f16 operands)
    bool range_fallback = false;           // yn_fold_bn found a folded GEMM weight outside the split's range (|w| >= 65504): same effect as exact_f32
    unsigned* range_host = nullptr;        // one word of pinned host memory: set by compact_kernel beside the negative counts (the range flag, out of band);
    unsigned* range_host_dev = nullptr;    // its device view.  While set, yn_infer / yn_pack_detections return YN_STATUS_RANGE; yn_range_status clears it
    unsigned* range_flags = nullptr;       // device uint[3]: [0] weights out of range (fold_pack_kernel), [1] an activation >= 65504 was split
                                           // (range_report, yn_device.h; read and cleared by yn_range_status), [2] the same as [0] for a yn_op_* call
    bool autotune = true;
    int force_pw_cfg = -1;                         // yn_set_pw_config (testing aid)
    int unit_chain = 1;                            // stride-1 ShuffleV2 units as one kernel each: 0 off, 1 where the map is large enough, 2 always (yn_unit_chain / YN_UNIT_CHAIN)
    bool pw_pipe = true;                           // pw_pipe_kernel among the pointwise candidates (yn_pw_pipe / YN_PW_PIPE=0)
    int chain_pipe = 1;                            // unit_pipe_kernel (the persistent per-unit walk): 0 never, 1 by the size rule, 2 always (yn_chain_pipe / YN_CHAIN_PIPE)
    int stage_fuse = 1;                            // all but the last stride-1 unit of a stage as ONE persistent launch (stage_pipe_kernel): 0 off, 1 from 256 tiles, 2 always (yn_stage_fuse / YN_STAGE_FUSE)
    int stage_pub_early = 0;                       // its tiles raise their ready flag right behind their stores (1) or under the next tile's depthwise phase (0, default: 608 x 608 stage 3 144 vs 150 us, 0.5x bs 128 101 vs 105, 416 bs 32 equal) (YN_STAGE_PUB)
    unsigned* stage_sync = nullptr;                // its queue heads / ready flags / exit count: behind the activation arena, zero between launches
    size_t stage_sync_bytes = 0;
    hipEvent_t tune_e0 = nullptr, tune_e1 = nullptr;
    std::vector<GraphEntry> graphs;
    bool profiling = false;
    std::vector<ProfRec> prof;
    std::vector<hipEvent_t> event_pool;
    size_t event_next = 0;
};

namespace {

int fail(yn_handle* h, const char* fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf; else g_create_error = buf;
    return 1;
}

// Every entry point runs with the handle's device current and restores the caller's afterwards (a handle created on
// cuda:1 may be called while cuda:0 is current: its hipMallocs, launches and hipFuncSetAttributes must not land there).
struct DevGuard {
    int prev = -1;
    explicit DevGuard(const yn_handle* h);
    ~DevGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
#define YN_ENTER(h) if (!(h)) return 1; DevGuard dev_guard_(h)

#define HIPCHK(h, expr)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess) return fail((h), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

DevGuard::DevGuard(const yn_handle* h)
{
    int cur = -1;
    if (hipGetDevice(&cur) == hipSuccess && cur != h->cfg.device && hipSetDevice(h->cfg.device) == hipSuccess) prev = cur;
}

// the f32-MFMA family runs every GEMM-shaped conv when the caller asked for it or when the folded weights do not fit the split
inline bool exact(const yn_handle* h) { return h->exact_f32 || h->range_fallback; }

const int STAGE_CH[4][3] = {{48, 96, 192}, {116, 232, 464}, {176, 352, 704}, {244, 488, 976}};
const int STAGE_REP[3] = {4, 8, 4};
const int NECK = 96;

void add_layer(yn_handle* h, const std::string& name, const std::string& conv, const std::string& bn,
               int kind, int cin, int cout, int stride, int has_bias, int act)
{
    Layer L;
    L.name = name; L.conv = conv; L.bn = bn; L.kind = kind; L.cin = cin; L.cout = cout;
    L.stride = stride; L.has_bias = has_bias; L.act = act;
    h->by_name[name] = (int)h->layers.size();
    h->by_conv[conv] = (int)h->layers.size();
    h->layers.push_back(L);
}

// the 77 convolutions, same naming as yolo_nano_amd/arch.py conv_specs()
void build_layers(yn_handle* h)
{
    char p[128], a[160], b[160];
    const int* sc = STAGE_CH[h->cfg.backbone];
    for (int i = 0; i < 3; ++i) h->stage_ch[i] = sc[i];
    add_layer(h, "stem", "backbone.conv1.0", "backbone.conv1.1", K_STEM, 3, 24, 2, 0, YN_ACT_RELU);
    int cin = 24;
    for (int si = 0; si < 3; ++si) {
        const int cout = sc[si], bf = cout / 2;
        for (int bi = 0; bi < STAGE_REP[si]; ++bi) {
            snprintf(p, sizeof p, "backbone.stage%d.%d", si + 2, bi);
            const std::string P = p;
            if (bi == 0) {
                add_layer(h, P + ".b1.dw", P + ".branch1.0", P + ".branch1.1", K_DW, cin, cin, 2, 0, YN_ACT_NONE);
                add_layer(h, P + ".b1.pw", P + ".branch1.2", P + ".branch1.3", K_PW, cin, bf, 1, 0, YN_ACT_RELU);
            }
            const int b2in = bi == 0 ? cin : bf;
            add_layer(h, P + ".b2.pw1", P + ".branch2.0", P + ".branch2.1", K_PW, b2in, bf, 1, 0, YN_ACT_RELU);
            add_layer(h, P + ".b2.dw", P + ".branch2.3", P + ".branch2.4", K_DW, bf, bf, bi == 0 ? 2 : 1, 0, YN_ACT_NONE);
            add_layer(h, P + ".b2.pw2", P + ".branch2.5", P + ".branch2.6", K_PW, bf, bf, 1, 0, YN_ACT_RELU);
        }
        cin = cout;
    }
    for (int i = 0; i < 3; ++i) {
        snprintf(a, sizeof a, "conv1x1_%d", i);
        add_layer(h, a, std::string(a) + ".convs.0", std::string(a) + ".convs.1", K_PW, sc[i], NECK, 1, 1, YN_ACT_LEAKY);
    }
    for (int i = 0; i < 4; ++i) {
        snprintf(a, sizeof a, "smooth_%d", i);
        add_layer(h, a, std::string(a) + ".convs.0", std::string(a) + ".convs.1", K_DENSE3, NECK, NECK, 1, 1, YN_ACT_LEAKY);
    }
    h->head_ch = h->cfg.num_anchors * (1 + h->cfg.num_classes + 4);
    for (int hd = 1; hd <= 3; ++hd) {
        for (int j = 0; j < 4; ++j) {
            snprintf(a, sizeof a, "head_det_%d.%d", hd, j);
            snprintf(b, sizeof b, "head_det_%d.%d.convs", hd, j);
            add_layer(h, a, std::string(b) + ".0", std::string(b) + ".1", (j & 1) ? K_PW : K_DW, NECK, NECK, 1, 1, YN_ACT_LEAKY);
        }
        snprintf(a, sizeof a, "head_det_%d.4", hd);
        add_layer(h, a, a, "", K_PW, NECK, h->head_ch, 1, 1, YN_ACT_NONE);
    }
}

int set_grid_info(yn_handle* h, int S)
{
    if (S <= 0 || (S % 32) != 0) return fail(h, "input_size %d is not a positive multiple of 32", S);
    GridInfo& g = h->grid;
    g.S = S; g.C = h->cfg.num_classes; g.A = h->cfg.num_anchors;
    int off = 0;
    for (int s = 0; s < 3; ++s) {
        const int w = S / (8 << s);
        g.w[s] = w; g.hw[s] = w * w; g.off[s] = off;
        off += w * w * g.A;
    }
    g.N = off;
    g.head_ld = g.A * (5 + g.C);
    for (int i = 0; i < 18; ++i) g.anchors[i] = h->cfg.anchors[i];
    return 0;
}

const Param* find_param(yn_handle* h, const std::string& key)
{
    auto it = h->params.find(key);
    return it == h->params.end() ? nullptr : &it->second;
}

// ---- arena ------------------------------------------------------------------------------------
size_t network_arena_bytes(yn_handle* h, int B, int S)
{
    // generous closed form: every buffer of run_network(), no reuse across stages
    const size_t px2 = (size_t)B * (S / 2) * (S / 2), px4 = px2 / 4;
    size_t fl = px2 * 24 + px4 * 24;
    int cin = 24;
    size_t cur = px4;
    for (int si = 0; si < 3; ++si) {
        const int C = h->stage_ch[si], bf = C / 2;
        const size_t po = cur / 4;
        fl += po * cin + po * bf + cur * bf + po * bf + 2 * po * C;
        cin = C; cur = po;
    }
    const size_t p3 = (size_t)B * (S / 8) * (S / 8), p4 = p3 / 4, p5 = p4 / 4;
    fl += (p3 + p4 + p5) * NECK * 2 + p4 * NECK;      // laterals, smoothed (p4 twice)
    fl += (p3 + p4 + p5) * NECK * 3;                   // head scratch, one set per (concurrent) head
    return fl * sizeof(float) + 192 * 256;           // (rounding + 16 bytes of slack per buffer: arena_take)
}

// every cached hipGraphExec_t refers to the buffers it was captured with: destroy them whenever those go away
void drop_train_graphs(yn_handle* h)
{
    if (!h->train_graphs.empty()) (void)hipStreamSynchronize(h->stream);
    for (TrainGraph& g : h->train_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    h->train_graphs.clear();
}

void drop_graphs(yn_handle* h)
{
    drop_train_graphs(h);
    if (!h->graphs.empty()) (void)hipStreamSynchronize(h->stream);     // a replay may still be in flight (the setters below reach here without draining)
    for (GraphEntry& g : h->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    h->graphs.clear();
}

int ensure_arena(yn_handle* h, int B, int S)
{
    const size_t need = network_arena_bytes(h, B, S);
    if (need <= h->arena_bytes) return 0;
    if (h->arena) { HIPCHK(h, hipStreamSynchronize(h->stream)); HIPCHK(h, hipFree(h->arena)); h->arena = nullptr; h->arena_bytes = 0; }
    // behind the activations: the sync words of stage_pipe_kernel (kernels_stage.hip) for the largest stage (stage 2: B (S/8)^2 rows, 32-row tiles,
    // up to YN_STAGE_MAX units).  Zeroed HERE, once, on the handle's stream; every launch leaves them zero.
    const size_t sync_bytes = (stage_sync_words((int)((size_t)B * (S / 8) * (S / 8) / 32 + 1), YN_STAGE_MAX) * sizeof(unsigned) + 255) & ~(size_t)255;
    HIPCHK(h, hipMalloc((void**)&h->arena, need + sync_bytes));
    h->arena_bytes = need;
    h->stage_sync = reinterpret_cast<unsigned*>(h->arena + need);
    h->stage_sync_bytes = sync_bytes;
    HIPCHK(h, hipMemsetAsync(h->stage_sync, 0, sync_bytes, h->stream));
    drop_graphs(h);
    return 0;
}

float* arena_take(yn_handle* h, size_t floats)
{
    // 16 bytes of slack behind every tensor: the LDS-DMA kernels fetch whole 16-byte pieces, and the last piece of a tensor's last row may run
    // up to 8 bytes past it (rows of 58 or 24 floats behind an 8-byte-aligned start) - never used, but it must be readable
    size_t bytes = (floats * sizeof(float) + 16 + 255) & ~(size_t)255;
    if (h->arena_used + bytes > h->arena_bytes) return nullptr;
    float* p = (float*)(h->arena + h->arena_used);
    h->arena_used += bytes;
    return p;
}

int ensure_post(yn_handle* h, int B, int N, int C)
{
    // worst case every candidate of an image falls into one class: that segment must fit resolve_segment()'s LDS mask
    if (N > nms_max_segment())
        return fail(h, "NMS over %d candidates per image exceeds the supported segment size %d", N, nms_max_segment());
    const size_t need = (size_t)B * N, need_seg = (size_t)B * (C + 1);
    const size_t m_stride = nms_matrix_words_per_image(N, C);
    const size_t need_m = m_stride * B;
    if (need > h->nms_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        void** ptrs[] = {(void**)&h->cand_boxes, (void**)&h->cand_scores, (void**)&h->cand_cls,
                         (void**)&h->nms.bucket, (void**)&h->nms.keep, (void**)&h->nms.sbox, (void**)&h->nms.bucket2, (void**)&h->nms.sbox2};
        for (void** q : ptrs) if (*q) { HIPCHK(h, hipFree(*q)); *q = nullptr; }
        HIPCHK(h, hipMalloc((void**)&h->cand_boxes, need * 4 * sizeof(float)));
        HIPCHK(h, hipMalloc((void**)&h->cand_scores, need * sizeof(float)));
        HIPCHK(h, hipMalloc((void**)&h->cand_cls, need * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.bucket, need * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.keep, need * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.sbox, need * 4 * sizeof(float)));
        HIPCHK(h, hipMalloc((void**)&h->nms.bucket2, need * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.sbox2, need * 4 * sizeof(float)));
        h->nms_cap = need;
        drop_graphs(h);
    }
    if (need_seg > h->nms_seg_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        void** ptrs[] = {(void**)&h->nms.seg_count, (void**)&h->nms.seg_off, (void**)&h->nms.tile_off, (void**)&h->nms.large_list,
                         (void**)&h->nms.seg_count2, (void**)&h->nms.tile_off2, (void**)&h->nms.seg_order, (void**)&h->nms.ctr, (void**)&h->nms.seg_sparse, (void**)&h->nms.work_off};
        for (void** q : ptrs) if (*q) { HIPCHK(h, hipFree(*q)); *q = nullptr; }
        HIPCHK(h, hipMalloc((void**)&h->nms.seg_count, need_seg * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.seg_off, need_seg * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.tile_off, need_seg * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.large_list, need_seg * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.seg_count2, need_seg * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.tile_off2, need_seg * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.seg_order, need_seg * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.seg_sparse, need_seg * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.work_off, need_seg * sizeof(int32_t)));
        HIPCHK(h, hipMalloc((void**)&h->nms.ctr, need_seg * sizeof(int32_t)));          // >= 2 * B ints
        HIPCHK(h, hipMemsetAsync(h->nms.ctr, 0, need_seg * sizeof(int32_t), h->stream));
        h->nms_seg_cap = need_seg;
        drop_graphs(h);
    }
    if (need_m > h->nms_m_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->nms.matrix) { HIPCHK(h, hipFree(h->nms.matrix)); h->nms.matrix = nullptr; }
        HIPCHK(h, hipMalloc(&h->nms.matrix, need_m * sizeof(unsigned long long)));
        h->nms_m_cap = need_m;
        drop_graphs(h);
    }
    const size_t need_ps = nms_pre_sync_words(B, N);
    if (need_ps > h->nms_ps_cap) {
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (h->nms.pre_sync) { HIPCHK(h, hipFree(h->nms.pre_sync)); h->nms.pre_sync = nullptr; }
        HIPCHK(h, hipMalloc(&h->nms.pre_sync, need_ps * sizeof(unsigned long long)));
        HIPCHK(h, hipMemsetAsync(h->nms.pre_sync, 0, need_ps * sizeof(unsigned long long), h->stream));
        h->nms_ps_cap = need_ps;
        drop_graphs(h);
    }
    h->nms.matrix_stride = m_stride;
    h->nms.large_cap = (N / 1024 + 1) < C ? (N / 1024 + 1) : C;       // at most N/1024 segments can exceed 1024 items
    return 0;
}

int ensure_heads(yn_handle* h, int B)
{
    const int hld = (h->head_ch + 3) & ~3;                  // internal heads: rows padded to 16 bytes => float4 stores
    const size_t need = (size_t)B * (h->grid.hw[0] + h->grid.hw[1] + h->grid.hw[2]) * hld;
    if (need <= h->heads_cap) {
        // re-derive the split for the current grid
        h->heads_int[1] = h->heads_int[0] + (size_t)B * h->grid.hw[0] * hld;
        h->heads_int[2] = h->heads_int[1] + (size_t)B * h->grid.hw[1] * hld;
        return 0;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (h->heads_int[0]) HIPCHK(h, hipFree(h->heads_int[0]));
    HIPCHK(h, hipMalloc((void**)&h->heads_int[0], need * sizeof(float) + 1024));
    h->heads_cap = need;
    h->heads_int[1] = h->heads_int[0] + (size_t)B * h->grid.hw[0] * hld;
    h->heads_int[2] = h->heads_int[1] + (size_t)B * h->grid.hw[1] * hld;
    drop_graphs(h);
    return 0;
}

// ---- profiling brackets ---------------------------------------------------------------------------
hipEvent_t take_event(yn_handle* h)
{
    if (h->event_next == h->event_pool.size()) {
        hipEvent_t e;
        (void)hipEventCreate(&e);
        h->event_pool.push_back(e);
    }
    return h->event_pool[h->event_next++];
}

struct Bracket {
    yn_handle* h;
    bool on;
    Bracket(yn_handle* h_, const std::string& name, double flops, double bytes) : h(h_), on(h_->profiling)
    {
        if (!on) return;
        ProfRec r;
        r.name = name; r.flops = flops; r.bytes = bytes;
        r.e0 = take_event(h); r.e1 = take_event(h);
        (void)hipEventRecord(r.e0, h->cur);
        h->prof.push_back(r);
    }
    void cancel() { if (on) { h->prof.pop_back(); h->event_next -= 2; on = false; } }
    ~Bracket()
    {
        if (!on) return;
        (void)hipEventRecord(h->prof.back().e1, h->cur);
        h->prof.back().kernel = last_kernel_name();
    }
};

// ---- layer launchers ------------------------------------------------------------------------------
const Layer& L(yn_handle* h, const std::string& name) { return h->layers[h->by_name.at(name)]; }

// Pick the fastest tile configuration for this (shape, strides) by timing every instantiated one on the
// handle's stream (all configurations are bit-identical, so this only affects speed).  Never runs during capture.
int tune_pw(yn_handle* h, GemmArgs a)
{
    static const int forced = getenv("YN_PW_FORCE_CFG") ? atoi(getenv("YN_PW_FORCE_CFG")) : -1;     // debugging / A-B runs
    if (forced >= 0) return forced;
    if (h->force_pw_cfg >= 0) return h->force_pw_cfg;
    if (!h->autotune) return -1;
    // the candidates: the split-f16 family for a layer that carries split packs, else the f32-MFMA family (each bit-identical inside)
    const int c_lo = a.Wsh ? pw_f32_config_count() : 0, c_hi = a.Wsh ? pw_config_count() - (h->pw_pipe ? 0 : 1) : pw_f32_config_count();   // (pw_pipe_kernel is the last index)
    // one table per process, shared by every handle (bench.py runs four per GPU: the layer shapes are timed once, not four times)
    const std::vector<int> key = {h->cfg.device, a.M, a.K, a.N, a.Npad, a.act, a.in_ld, a.in_off, a.out_ld, a.out_off, a.pass ? 1 : 0, a.Wsh ? 1 : 0, h->pw_pipe ? 1 : 0};
    {
        std::lock_guard<std::mutex> lk(g_tune_mutex);
        auto it = g_pw_tuned.find(key);
        if (it != g_pw_tuned.end()) return it->second;
    }
    if (h->profiling) return -1;
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(h->cur, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return -1;
    if (!h->tune_e0) { (void)hipEventCreate(&h->tune_e0); (void)hipEventCreate(&h->tune_e1); }
    int best = -1;
    float best_ms = 1e30f;
    // two rounds over all candidates, minimum per candidate; every bracket spans >= ~300 us of launches (5...40 of them): a single
    // short bracket is noisy enough (clock ramp, a neighbour stream's kernel) to pick a tile 1.5x slower than the best one, and the
    // named workloads then moved by +-10 % from run to run
    int reps = 5;
    {
        a.cfg = c_lo;
        launch_pw(a, h->cur);
        (void)hipEventRecord(h->tune_e0, h->cur);
        launch_pw(a, h->cur); launch_pw(a, h->cur);
        (void)hipEventRecord(h->tune_e1, h->cur);
        float ms2 = 0.0f;
        if (hipEventSynchronize(h->tune_e1) != hipSuccess) return -1;
        (void)hipEventElapsedTime(&ms2, h->tune_e0, h->tune_e1);
        const float per = ms2 > 0.0f ? ms2 * 0.5f : 0.02f;
        reps = (int)(0.3f / per);
        reps = reps < 5 ? 5 : (reps > 40 ? 40 : reps);
    }
    for (int round = 0; round < 2; ++round) {
        for (int c = c_lo; c < c_hi; ++c) {
            a.cfg = c;
            launch_pw(a, h->cur);                           // warm-up
            (void)hipEventRecord(h->tune_e0, h->cur);
            for (int r = 0; r < reps; ++r) launch_pw(a, h->cur);
            (void)hipEventRecord(h->tune_e1, h->cur);
            if (hipEventSynchronize(h->tune_e1) != hipSuccess) return -1;
            float ms = 0.0f;
            (void)hipEventElapsedTime(&ms, h->tune_e0, h->tune_e1);
            if (ms < best_ms) { best_ms = ms; best = c; }
        }
    }
    if (hipGetLastError() != hipSuccess) return -1;
    {
        std::lock_guard<std::mutex> lk(g_tune_mutex);
        g_pw_tuned[key] = best;
    }
    return best;
}

// Timing ablation (tools/ablate.sh): YN_DBG_SKIP_LAYERS=<substr,substr,...> drops the launches of every layer whose name contains one of
// the substrings, from the (YN_DBG_SKIP_AFTER, default 8)-th network pass of the handle on - the activation arena then still holds the
// layer's output of the earlier passes on the same input, so everything downstream (NMS included) does its normal work.
static bool dbg_skip(yn_handle* h, const std::string& name)
{
    static const char* pat = getenv("YN_DBG_SKIP_LAYERS");
    if (!pat) return false;
    static const int after = getenv("YN_DBG_SKIP_AFTER") ? atoi(getenv("YN_DBG_SKIP_AFTER")) : 8;
    if (h->net_passes <= after) return false;
    std::string p(pat);
    size_t i = 0;
    while (i <= p.size()) {
        const size_t j = p.find(',', i);
        const std::string tok = p.substr(i, j == std::string::npos ? std::string::npos : j - i);
        if (!tok.empty() && name.find(tok) != std::string::npos) return true;
        if (j == std::string::npos) break;
        i = j + 1;
    }
    return false;
}

GemmArgs pw_args(yn_handle* h, const Layer& l, const float* in, int in_ld, int in_off, long M,
                 float* out, int out_ld, int out_off, const float* pass, int pass_ld, int pass_off, int n_store = 0)
{
    GemmArgs a{};
    a.in = in; a.in_ld = in_ld; a.in_off = in_off;
    a.Wp = l.w_packed; a.bias = l.b_packed;
    a.out = out; a.out_ld = out_ld; a.out_off = out_off;
    a.pass = pass; a.pass_ld = pass_ld; a.pass_off = pass_off;
    a.M = (int)M; a.K = l.cin; a.N = l.cout; a.Npad = l.Npad; a.act = l.act;
    a.in_slack = (h->arena && (const char*)in >= h->arena && (const char*)in < h->arena + h->arena_bytes) ? 16 : 0;
    if (n_store > l.cout && n_store <= l.Npad) a.N = n_store;     // padded output row: the extra (zero-weight) columns are stored too
    if (!exact(h)) { a.Wsh = l.ws_hi; a.Wsl = l.ws_lo; }          // split-f16 MFMA family (fp32-class); exact_f32 / range fallback: the f32-MFMA kernels
    a.ovf = h->range_flags ? h->range_flags + 1 : nullptr;
    a.cfg = -1;
    return a;
}

void run_pw(yn_handle* h, const Layer& l, const float* in, int in_ld, int in_off, long M,
            float* out, int out_ld, int out_off, const float* pass, int pass_ld, int pass_off, int n_store = 0)
{
    GemmArgs a = pw_args(h, l, in, in_ld, in_off, M, out, out_ld, out_off, pass, pass_ld, pass_off, n_store);
    a.cfg = tune_pw(h, a);
    if (dbg_skip(h, l.name)) return;
    Bracket br(h, l.name, 2.0 * M * l.cin * l.cout,
               4.0 * (M * (double)(l.cin + l.cout + (pass ? 2 * l.cout : 0)) + (double)l.cin * l.cout));
    launch_pw(a, h->cur);
}

void run_dw(yn_handle* h, const Layer& l, const float* in, int in_ld, int in_off, int B, int H, int W,
            float* out, int out_ld, int out_off)
{
    DwArgs a{};
    a.in = in; a.in_ld = in_ld; a.in_off = in_off; a.w = l.w_packed; a.bias = l.b_packed;
    a.out = out; a.out_ld = out_ld; a.out_off = out_off;
    a.B = B; a.H = H; a.W = W; a.C = l.cout; a.stride = l.stride; a.act = l.act;
    const double Mi = (double)B * H * W, Mo = (double)B * ((H - 1) / l.stride + 1) * ((W - 1) / l.stride + 1);
    if (dbg_skip(h, l.name)) return;
    Bracket br(h, l.name, 2.0 * Mo * 9 * l.cout, 4.0 * (Mi + Mo) * l.cout);
    launch_dw(a, h->cur);
}

// stride-1 ShuffleV2 unit (backbone/shufflenetv2.py:70-72): x [B,H,W,C] -> out [B,H,W,C] as pointwise -> depthwise ->
// pointwise with the concat+shuffle epilogue (t1, t2 scratch); the one-kernel-per-unit form is run_unit_chain below
void run_dwpw(yn_handle* h, const Layer& dw, const Layer& pw, const float* in, int in_ld, int in_off, int B, int H, int W,
              float* tmp, float* out, int out_ld, int out_off, const float* pass, int pass_ld, int pass_off);
void run_unit(yn_handle* h, const std::string& P, const float* x, int B, int H, int W, float* out, float* t1, float* t2)
{
    const Layer& pw1 = L(h, P + ".b2.pw1");
    const Layer& dw = L(h, P + ".b2.dw");
    const Layer& pw2 = L(h, P + ".b2.pw2");
    const int bf = pw1.cout, C = 2 * bf;
    const long M = (long)B * H * W;
    run_pw(h, pw1, x, C, bf, M, t1, bf, 0, nullptr, 0, 0);
    run_dwpw(h, dw, pw2, t1, bf, 0, B, H, W, t2, out, C, 0, x, C, 0);
}

// Units 1..R-1 of a stage as pw1(unit 1) + one unit_chain_kernel per unit.  oA = the stride-2 block's output [M][C] (unit 1's
// input), oB = a second [M][C] buffer (holds the two [M][bf] pass-through halves in flight), tA / tB = [M][bf] scratch.
// Returns 0 (nothing launched) when the chain is off, the map is too small or no instantiated tile covers the shape; 1 when
// the stage ran, *result = final [M][C]; -1 on an error (latched in the handle).
// dry: only answer whether the stage will run as a chain (nothing launched).  t1_ready: unit 1's pw1 output is already in tA (the stride-2
// unit's kernel computed it: down2_kernel's last phase)
int run_unit_chain(yn_handle* h, int stage, int R, float* oA, int B, int H, int W, int C, float* oB, float* tA, float* tB, float** result,
                   bool dry = false, bool t1_ready = false)
{
    if (!h->unit_chain) return 0;
    const int bf = C / 2;
    const long M = (long)B * H * W;
    // Stages 2 and 3 (bf <= 128) chain at every size: on one image's maps the 32-row tiles of unit_chain2_kernel run a unit in 12 - 16 us
    // against 8 + 6 + 8 for its three kernels (bs = 1: 0.620 -> 0.585 ms at 416 x 416, 0.757 -> 0.719 ms at 608 x 608).  Stage 4 (bf = 232) chained
    // from M = 4 096 pixels only while the chain kernel staged its weights through LDS (eight chunk rounds per GEMM: 32 + 30 + 19 us against 3 x 29 on
    // one image).  With the weights register-direct (round 4) the chain costs one image what its three launches did (0.5245 vs 0.5227 ms at 416,
    // 0.653-0.660 either way at 608; bs 2 / 4 / 8 / 16: 0.554 / 0.622 / 0.714 / 0.847 vs 0.548 / 0.617 / 0.718 / 0.850 ms) and is six launches
    // less: it chains at every size now (YN_CHAIN_MIN4=<M> restores a threshold).
    static const long min4 = getenv("YN_CHAIN_MIN4") ? atol(getenv("YN_CHAIN_MIN4")) : 0;       // A/B: smallest M that chains the bf > 128 stage
    if (h->unit_chain != 2 && bf > 128 && M < min4) return 0;
    char nm[96];
    auto name = [&](int bi) { snprintf(nm, sizeof nm, "backbone.stage%d.%d", stage, bi); return std::string(nm); };
    {
        const Layer& pw1 = L(h, name(1) + ".b2.pw1");
        const Layer& dw = L(h, name(1) + ".b2.dw");
        if (pw1.cin != bf || pw1.cout != bf || dw.stride != 1) return 0;
        // every unit of the stage must be covered by an instantiated tile BEFORE anything is launched: the chain consumes its
        // input buffers, so there is no falling back half way (unit 1 reads x1 with ld = C, the others with ld = bf)
        for (int first = 0; first < 2; ++first) {
            ChainArgs q{};
            q.t1_ld = bf; q.x1_ld = first ? C : bf; q.out_ld = bf; q.bf = bf; q.Npad = L(h, name(1) + ".b2.pw2").Npad; q.M = (int)M;
            if (!exact(h)) q.Ws2h = L(h, name(1) + ".b2.pw2").ws_hi;
            if (!unit_chain_covers(q)) return 0;
            q.out_ld = C;
            if (!unit_chain_covers(q)) return 0;
        }
        if (dry) return 1;
        if (!t1_ready) run_pw(h, pw1, oA, C, bf, M, tA, bf, 0, nullptr, 0, 0);
    }
    float* pbuf[2] = {oB, oB + (size_t)M * bf};
    const float* x1 = oA;
    int x1_ld = C;
    float* final_out = R > 2 ? oA : oB;
    std::vector<ChainArgs> ua;
    for (int bi = 1; bi < R; ++bi) {
        const std::string P = name(bi);
        const bool last = bi == R - 1;
        const Layer& dw = L(h, P + ".b2.dw");
        const Layer& pw2 = L(h, P + ".b2.pw2");
        ChainArgs a{};
        a.t1 = tA; a.t1_ld = bf; a.t1_off = 0;
        a.x1 = x1; a.x1_ld = x1_ld; a.x1_off = 0;
        a.wdw = dw.w_packed; a.bdw = dw.b_packed; a.dw_act = dw.act;
        a.Wp2 = pw2.w_packed; a.b2 = pw2.b_packed; a.act2 = pw2.act;
        if (!exact(h)) { a.Ws2h = pw2.ws_hi; a.Ws2l = pw2.ws_lo; }
        if (!last) {
            const Layer& pw1n = L(h, name(bi + 1) + ".b2.pw1");
            a.Wp1n = pw1n.w_packed; a.b1n = pw1n.b_packed; a.act1n = pw1n.act;
            if (!exact(h)) { a.Ws1h = pw1n.ws_hi; a.Ws1l = pw1n.ws_lo; }
            a.out = pbuf[(bi - 1) & 1]; a.out_ld = bf; a.t1n = tB;
        } else {
            a.out = final_out; a.out_ld = C;
        }
        a.B = B; a.H = H; a.W = W; a.bf = bf; a.Npad = pw2.Npad; a.M = (int)M;
        a.pipe_mode = h->chain_pipe == 1 ? 0 : (h->chain_pipe == 0 ? 1 : 2);
        a.ovf = h->range_flags ? h->range_flags + 1 : nullptr;
        ua.push_back(a);
        x1 = a.out; x1_ld = bf;
        float* tmp = tA; tA = tB; tB = tmp;
    }
    // All but the last unit as ONE persistent launch (stage_pipe_kernel, kernels_stage.hip) where a form exists: the conditions of
    // unit_pipe_kernel (split-f16 family, ReLU pointwise / linear depthwise convs, dense inputs) + channel quads + enough tiles to walk
    int staged = 0;
    if (h->stage_fuse && !exact(h) && R - 2 >= 2 && R - 2 <= YN_STAGE_MAX && h->stage_sync) {
        StageArgs sa{};
        bool ok = true;
        double fl = 0, by = 0;
        for (int i = 0; i < R - 2 && ok; ++i) {
            const ChainArgs& q = ua[i];
            ok = q.Ws2h && q.Ws1h && q.dw_act == 0 && q.act2 == 1 && q.act1n == 1 && q.t1_ld == bf && q.t1_off == 0 && q.x1_off == 0 && (q.x1_ld & 3) == 0 &&
                 q.out_ld == bf && q.Npad == ((bf + 31) & ~31);
            StageUnit& u = sa.u[i];
            u.t1 = q.t1; u.x1 = q.x1; u.x1_ld = q.x1_ld; u.wdw = q.wdw; u.bdw = q.bdw; u.b2 = q.b2; u.b1n = q.b1n;
            u.Ws2h = q.Ws2h; u.Ws2l = q.Ws2l; u.Ws1h = q.Ws1h; u.Ws1l = q.Ws1l; u.out = q.out; u.t1n = q.t1n;
            fl += 2.0 * M * bf * (9.0 + 2.0 * bf);
            by += 4.0 * (4.0 * M * bf + 2.0 * bf * bf + 10.0 * bf);
        }
        sa.nunits = R - 2; sa.M = (int)M; sa.H = H; sa.W = W;
        sa.inv_w = 1.0f / (float)W; sa.inv_h = 1.0f / (float)H;
        sa.ovf = h->range_flags ? h->range_flags + 1 : nullptr;
        sa.sync = h->stage_sync;
        const int min_tiles = h->stage_fuse == 2 ? 1 : 256;
        if (ok && launch_stage_pipe(sa, bf, h->stage_pub_early, min_tiles, h->stage_sync_bytes, h->cur, true)) {
            snprintf(nm, sizeof nm, "backbone.stage%d.1-%d.dw+pw2+pw1n", stage, R - 2);
            const std::string sname = nm;
            if (!dbg_skip(h, sname) && !dbg_skip(h, name(1) + ".chain")) {
                Bracket br(h, sname, fl, by);
                (void)launch_stage_pipe(sa, bf, h->stage_pub_early, min_tiles, h->stage_sync_bytes, h->cur);
            }
            staged = R - 2;
        }
    }
    for (int bi = 1 + staged; bi < R; ++bi) {
        const std::string P = name(bi);
        const bool last = bi == R - 1;
        const ChainArgs& a = ua[bi - 1];
        Bracket br(h, P + (last ? ".dw+pw2" : ".dw+pw2+pw1n"), 2.0 * M * bf * (9.0 + bf + (last ? 0.0 : (double)bf)),
                   4.0 * (4.0 * M * bf + (last ? 1.0 : 2.0) * bf * bf + 10.0 * bf));
        if (dbg_skip(h, P + ".chain")) { br.cancel(); continue; }
        if (!launch_unit_chain(a, h->cur)) {                 // cannot happen after the coverage check above
            br.cancel();
            fail(h, "unit chain: no tile for stage %d unit %d after the coverage check", stage, bi);
            return -1;                                         // never fall back here: the chain has consumed its input buffers
        }
    }
    *result = final_out;
    return 1;
}

void run_dwpw(yn_handle* h, const Layer& dw, const Layer& pw, const float* in, int in_ld, int in_off, int B, int H, int W,
              float* tmp, float* out, int out_ld, int out_off, const float* pass, int pass_ld, int pass_off)
{
    const int Ho = (H - 1) / dw.stride + 1, Wo = (W - 1) / dw.stride + 1;
    const long M = (long)B * Ho * Wo;                       // output pixels
    run_dw(h, dw, in, in_ld, in_off, B, H, W, tmp, dw.cout, 0);
    run_pw(h, pw, tmp, dw.cout, 0, M, out, out_ld, out_off, pass, pass_ld, pass_off);
}

void run_c3(yn_handle* h, const Layer& l, const float* in, const float* in2, int resample, int B, int H, int W, float* out)
{
    GemmArgs a{};
    a.in = in; a.in_ld = l.cin; a.in_off = 0; a.in2 = in2; a.resample = resample; a.H = H; a.W = W;
    a.Wp = l.w_packed; a.bias = l.b_packed; a.out = out; a.out_ld = l.cout; a.out_off = 0;
    a.M = B * H * W; a.K = l.cin; a.N = l.cout; a.Npad = l.Npad; a.act = l.act;
    a.cfg = -1;
    if (!exact(h)) { a.Wsh = l.ws_hi; a.Wsl = l.ws_lo; }         // split-f16 MFMA path (fp32-class); exact_f32 / range fallback: the f32-MFMA kernel
    a.ovf = h->range_flags ? h->range_flags + 1 : nullptr;
    if (dbg_skip(h, l.name)) return;
    const double M = (double)a.M;
    const double in2px = resample == 1 ? M / 4 : (resample == 2 ? M * 4 : 0);
    Bracket br(h, l.name, 2.0 * M * 9 * l.cin * l.cout, 4.0 * ((M + in2px) * l.cin + M * l.cout + 9.0 * l.cin * l.cout));
    launch_conv3x3(a, h->cur);
}

// ---- grouped launches (Group<>, yn_internal.h): the same layer of the three detection heads / the three laterals as ONE launch ----
bool run_pw_group(yn_handle* h, const Layer* const l[], GemmArgs a[], int n, const char* name)
{
    // the tile configuration is timed for the GROUP (one launch of every split configuration, as tune_pw does for a single layer): the
    // best tile of the largest problem can leave a small-M / long-K member with a handful of serial workgroups (the laterals: K = 464 on
    // 5 408 pixels rode for 57 us on the 128-row tile of the 86 528-pixel member).  Every split configuration gives the same bits.
    a[0].cfg = -1;
    if (h->autotune && h->force_pw_cfg < 0) {
        std::vector<int> key = {h->cfg.device, -n};
        for (int p = 0; p < n; ++p) { key.push_back(a[p].M); key.push_back(a[p].K); key.push_back(a[p].N); key.push_back(a[p].in_ld); key.push_back(a[p].out_ld); }
        bool found = false;
        {
            std::lock_guard<std::mutex> lk(g_tune_mutex);
            auto it = g_pw_tuned.find(key);
            if (it != g_pw_tuned.end()) { a[0].cfg = it->second; found = true; }
        }
        hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
        if (!found && !h->profiling && hipStreamIsCapturing(h->cur, &st) == hipSuccess && st == hipStreamCaptureStatusNone) {
            if (!h->tune_e0) { (void)hipEventCreate(&h->tune_e0); (void)hipEventCreate(&h->tune_e1); }
            int best = -1;
            float best_ms = 1e30f;
            for (int round = 0; round < 2; ++round)
                for (int c = pw_f32_config_count(); c < pw_config_count(); ++c) {
                    if (!launch_pw_group(a, n, c, h->cur)) continue;                 // warm-up
                    (void)hipEventRecord(h->tune_e0, h->cur);
                    for (int r = 0; r < 5; ++r) (void)launch_pw_group(a, n, c, h->cur);
                    (void)hipEventRecord(h->tune_e1, h->cur);
                    if (hipEventSynchronize(h->tune_e1) != hipSuccess) { best = -1; break; }
                    float ms = 0.0f;
                    (void)hipEventElapsedTime(&ms, h->tune_e0, h->tune_e1);
                    if (ms < best_ms) { best_ms = ms; best = c; }
                }
            if (best >= 0) {
                std::lock_guard<std::mutex> lk(g_tune_mutex);
                g_pw_tuned[key] = best;
                a[0].cfg = best;
            }
        }
    } else if (h->force_pw_cfg >= 0) a[0].cfg = h->force_pw_cfg;
    if (dbg_skip(h, name)) return true;
    double fl = 0, by = 0;
    for (int p = 0; p < n; ++p) {
        fl += 2.0 * a[p].M * l[p]->cin * l[p]->cout;
        by += 4.0 * (a[p].M * (double)(l[p]->cin + l[p]->cout) + (double)l[p]->cin * l[p]->cout);
    }
    Bracket br(h, name, fl, by);
    if (launch_pw_group(a, n, a[0].cfg, h->cur)) return true;
    br.cancel();
    return false;
}

void run_dw_group(yn_handle* h, const Layer* const l[], DwArgs a[], int n, const char* name)
{
    double fl = 0, by = 0;
    for (int p = 0; p < n; ++p) {
        const double M = (double)a[p].B * a[p].H * a[p].W;
        fl += 2.0 * M * 9 * l[p]->cout;
        by += 4.0 * 2 * M * l[p]->cout;
    }
    if (dbg_skip(h, name)) return;
    Bracket br(h, name, fl, by);
    launch_dw_group(a, n, h->cur);
}

// ---- fork / join of independent kernel chains onto side streams (also legal inside stream capture) ----
hipEvent_t fj_event(yn_handle* h)
{
    if (h->fj_events.size() < 32) {
        hipEvent_t e;
        (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
        h->fj_events.push_back(e);
        return e;
    }
    return h->fj_events[h->fj_next++ % h->fj_events.size()];
}
bool side_ok(yn_handle* h)
{
    if (!h->multi_stream || h->profiling || h->stream == nullptr) return false;
    for (int k = 0; k < 2; ++k)
        if (!h->side[k] && hipStreamCreateWithFlags(&h->side[k], hipStreamNonBlocking) != hipSuccess) return false;
    return true;
}
// side stream k picks up after everything issued so far on the main stream; subsequent run_* target it
void fork_to(yn_handle* h, int k)
{
    if (!side_ok(h)) return;
    hipEvent_t e = fj_event(h);
    (void)hipEventRecord(e, h->stream);
    (void)hipStreamWaitEvent(h->side[k], e, 0);
    h->cur = h->side[k];
}
void back_to_main(yn_handle* h) { h->cur = h->stream; }
// the main stream waits for everything issued so far on side stream k
void join_from(yn_handle* h, int k)
{
    if (!h->side[k] || !side_ok(h)) return;
    hipEvent_t e = fj_event(h);
    (void)hipEventRecord(e, h->side[k]);
    (void)hipStreamWaitEvent(h->stream, e, 0);
}

static GemmArgs head_final_args(yn_handle* h, const Layer& l, const float* in, long M)
{
    GemmArgs a{};
    a.in = in; a.in_ld = NECK; a.in_off = 0;
    a.Wp = l.w_packed; a.bias = l.b_packed;
    a.M = (int)M; a.K = l.cin; a.N = l.cout; a.Npad = l.Npad; a.act = l.act;
    a.Wsh = l.ws_hi; a.Wsl = l.ws_lo;
    a.ovf = h->range_flags ? h->range_flags + 1 : nullptr;
    a.cfg = -1;
    return a;
}

// The network: x NCHW [B,3,S,S] -> three NHWC head tensors.
// fuse_decode: the last conv of each head runs as head_decode_kernel (candidates written straight into h->cand_*, raw heads not stored);
// *fused reports whether all three heads took that path (else the caller runs decode_kernel on the raw heads).
int run_network(yn_handle* h, const float* x, int B, float* const heads[3], int head_ld, bool fuse_decode = false, bool* fused = nullptr)
{
    const int S = h->grid.S;
    h->arena_used = 0;
    h->cur = h->stream;
    ++h->net_passes;
#define TAKE(var, floats)                                                                   \
    float* var = arena_take(h, (size_t)(floats));                                           \
    if (!var) return fail(h, "activation arena exhausted (%zu bytes)", h->arena_bytes)
    const int H1 = S / 2, H2 = S / 4;
    TAKE(a0, (size_t)B * H1 * H1 * 24);
    TAKE(a1, (size_t)B * H2 * H2 * 24);
    {
        // stem conv + maxpool in one kernel: the [B,S/2,S/2,24] conv activation stays in LDS (a0 is unused)
        const Layer& l = L(h, "stem");
        const double Mo = (double)B * H1 * H1;
        Bracket br(h, "stem+maxpool", 2.0 * Mo * 27 * 24, 4.0 * ((double)B * 3 * S * S + (double)B * H2 * H2 * 24));
        if (!dbg_skip(h, "stem")) launch_stem_pool(x, B, S, S, l.w_packed, l.b_packed, l.cout, l.act, a1, h->cur);
    }
    const float* cur = a1;
    int curC = 24, curH = H2;
    const float* cfeat[3];
    char nm[96];
    for (int si = 0; si < 3; ++si) {
        const int C = h->stage_ch[si], bf = C / 2, Ho = curH / 2;
        const long Mi = (long)B * curH * curH, Mo = (long)B * Ho * Ho;
        TAKE(tdw1, Mo * curC);
        TAKE(tb1, Mo * bf);
        TAKE(t1, Mi * bf);
        TAKE(t2, Mo * bf);
        TAKE(oA, Mo * C);
        TAKE(oB, Mo * C);
        snprintf(nm, sizeof nm, "backbone.stage%d.0", si + 2);
        const std::string P0 = nm;
        bool down_chained = false;                          // the stride-2 unit's kernel has also produced unit 1's pw1 output (in t2)
        // stride-2 block: backbone/shufflenetv2.py:73-74
        {
            // the whole unit as ONE kernel where its tile fits (cin <= 32, bf <= 64: stage 2, whose pw1 output is the largest tensor of the
            // network): branch 2 = pw1 -> depthwise s2 -> pw2, branch 1 = depthwise s2 -> pw, concat + shuffle in the store
            const Layer &lp1 = L(h, P0 + ".b2.pw1"), &ldw = L(h, P0 + ".b2.dw"), &lp2 = L(h, P0 + ".b2.pw2");
            const Layer &l1d = L(h, P0 + ".b1.dw"), &l1p = L(h, P0 + ".b1.pw");
            DownArgs d{};
            d.x = cur; d.cin = curC;
            d.W1h = lp1.ws_hi; d.W1l = lp1.ws_lo; d.b1 = lp1.b_packed; d.act1 = lp1.act; d.Npad1 = lp1.Npad;
            d.wdw = ldw.w_packed; d.bdw = ldw.b_packed; d.dw_act = ldw.act;
            d.W2h = lp2.ws_hi; d.W2l = lp2.ws_lo; d.b2 = lp2.b_packed; d.act2 = lp2.act; d.Npad2 = lp2.Npad;
            d.pass = nullptr;
            d.wdw1 = l1d.w_packed; d.bdw1 = l1d.b_packed; d.dw1_act = l1d.act;
            d.W3h = l1p.ws_hi; d.W3l = l1p.ws_lo; d.b3 = l1p.b_packed; d.act3 = l1p.act; d.Npad3 = l1p.Npad;
            d.out = oA; d.B = B; d.H = curH; d.W = curH; d.bf = bf;
            d.ovf = h->range_flags ? h->range_flags + 1 : nullptr;
            const bool use_down = h->down_fuse && !exact(h) && lp1.cin == curC && lp1.cout == bf && lp2.cin == bf && lp2.cout == bf && ldw.stride == 2 &&
                                  l1d.stride == 2 && l1d.cout == curC && l1p.cin == curC && l1p.cout == bf && down_unit_covers(d);
            static const int down_b1 = getenv("YN_DOWN_B1") ? atoi(getenv("YN_DOWN_B1")) : 1;      // 0: branch 1 as its own two kernels (A/B runs)
            static const int down2_env = getenv("YN_DOWN2") ? atoi(getenv("YN_DOWN2")) : 1;        // 0: the wide units (stages 3 / 4) as five launches (A/B runs)
            Down2Args d2{};
            d2.x = cur; d2.cin = curC; d2.y1 = t1;
            d2.wdw = ldw.w_packed; d2.bdw = ldw.b_packed; d2.dw_act = ldw.act;
            d2.W2h = lp2.ws_hi; d2.W2l = lp2.ws_lo; d2.b2 = lp2.b_packed; d2.act2 = lp2.act;
            d2.wdw1 = l1d.w_packed; d2.bdw1 = l1d.b_packed; d2.dw1_act = l1d.act;
            d2.W3h = l1p.ws_hi; d2.W3l = l1p.ws_lo; d2.b3 = l1p.b_packed; d2.act3 = l1p.act;
            d2.out = oA; d2.B = B; d2.H = curH; d2.W = curH; d2.bf = bf; d2.Npad = lp2.Npad;
            d2.ovf = h->range_flags ? h->range_flags + 1 : nullptr;
            const bool use_down2 = !use_down && h->down_fuse && down2_env && !exact(h) && lp1.ws_hi && lp1.cin == curC && lp1.cout == bf && lp2.cin == bf && lp2.cout == bf &&
                                   ldw.stride == 2 && ldw.cout == bf && l1d.stride == 2 && l1d.cout == curC && l1p.cin == curC && l1p.cout == bf && l1p.Npad == lp2.Npad &&
                                   down2_covers(d2);
            // when the stride-1 units behind it run as a chain, the kernel also computes unit 1's pw1 (on channels [bf, 2bf) of its own output)
            // ... in the launch-latency regime only (at most 64 tiles of 32 pixels: one to a few images - one 608 x 608 image saves two 10-15 us
            // launches).  Beyond that the extra k-steps stream their weights at the CU's L1 rate (each 32-row workgroup pulls the whole
            // matrix): 42 us against 30 + 13 at stage 3 (676 tiles), 58 against 34 + 17 at stage 4 (169 tiles) of a 32-image batch.
            static const int down2_next = getenv("YN_DOWN2_NEXT") ? atoi(getenv("YN_DOWN2_NEXT")) : 1;     // 0: never, 2: always (A/B runs, tests)
            const bool next_pays = down2_next == 2 || (down2_next == 1 && Mo <= 32 * 64);
            if (use_down2 && next_pays && STAGE_REP[si] > 1 && run_unit_chain(h, si + 2, STAGE_REP[si], oA, B, Ho, Ho, C, oB, t2, t1, nullptr, true) == 1) {
                snprintf(nm, sizeof nm, "backbone.stage%d.1.b2.pw1", si + 2);
                const Layer& l1n = L(h, nm);
                if (l1n.ws_hi && l1n.cin == bf && l1n.cout == bf && l1n.Npad == lp2.Npad) {
                    d2.W1nh = l1n.ws_hi; d2.W1nl = l1n.ws_lo; d2.b1n = l1n.b_packed; d2.act1n = l1n.act; d2.t1n = t2;
                    down_chained = true;
                }
            }
            if (use_down && !down_b1) {
                run_dw(h, l1d, cur, curC, 0, B, curH, curH, tdw1, curC, 0);
                run_pw(h, l1p, tdw1, curC, 0, Mo, tb1, bf, 0, nullptr, 0, 0);
                d.pass = tb1;
                Bracket br(h, P0 + ".b2", 2.0 * (Mi * curC * bf + Mo * bf * (9.0 + bf)), 4.0 * (Mi * (double)curC + 3.0 * Mo * bf + (double)curC * bf + (double)bf * bf));
                launch_down_unit(d, h->cur);
            } else if (use_down) {
                if (!dbg_skip(h, P0 + ".unit")) {
                    Bracket br(h, P0 + ".unit", 2.0 * (Mi * curC * bf + Mo * bf * (9.0 + bf) + Mo * curC * (9.0 + bf)),
                               4.0 * (Mi * (double)curC + 2.0 * Mo * bf + 2.0 * (double)curC * bf + (double)bf * bf));
                    launch_down_unit(d, h->cur);
                }
            } else if (use_down2) {
                // stages 3 / 4 (cin = bf = 116 / 232): pw1 as a GEMM launch, everything behind it - both depthwise convs, both pointwise convs,
                // concat + shuffle - as one kernel (down2_kernel): two launches instead of five
                run_pw(h, lp1, cur, curC, 0, Mi, t1, bf, 0, nullptr, 0, 0);
                if (!dbg_skip(h, P0 + ".tail")) {
                    const double nx = down_chained ? 1.0 : 0.0;
                    Bracket br(h, P0 + (down_chained ? ".dw+pw2|b1+pw1n" : ".dw+pw2|b1"), 2.0 * (Mo * bf * (9.0 + bf) + Mo * curC * (9.0 + bf) + nx * Mo * bf * bf),
                               4.0 * (Mi * (double)(curC + bf) + (2.0 + nx) * Mo * bf + (double)curC * bf + (1.0 + nx) * bf * bf));
                    launch_down2(d2, h->cur);
                }
            } else {
        fork_to(h, 0);                                      // branch1 and branch2 only meet in the fused cat+shuffle
        {
            run_dw(h, L(h, P0 + ".b1.dw"), cur, curC, 0, B, curH, curH, tdw1, curC, 0);
            run_pw(h, L(h, P0 + ".b1.pw"), tdw1, curC, 0, Mo, tb1, bf, 0, nullptr, 0, 0);
            back_to_main(h);
            run_pw(h, L(h, P0 + ".b2.pw1"), cur, curC, 0, Mi, t1, bf, 0, nullptr, 0, 0);
            run_dw(h, L(h, P0 + ".b2.dw"), t1, bf, 0, B, curH, curH, t2, bf, 0);
            join_from(h, 0);
            run_pw(h, L(h, P0 + ".b2.pw2"), t2, bf, 0, Mo, oA, C, 0, tb1, bf, 0);     // cat + shuffle fused
        }
            }
        }
        float* o_cur = oA;
        float* o_nxt = oB;
        // stride-1 blocks (backbone/shufflenetv2.py:70-72; x1 = ch [0,bf) passes through, x2 = ch [bf,C)): one kernel per
        // unit, cut at the depthwise conv (unit_chain_kernel), after the first unit's pw1; else three kernels per unit
        const int R = STAGE_REP[si];
        const int chained = R > 1 ? (down_chained ? run_unit_chain(h, si + 2, R, o_cur, B, Ho, Ho, C, o_nxt, t2, t1, &o_cur, false, true)
                                                  : run_unit_chain(h, si + 2, R, o_cur, B, Ho, Ho, C, o_nxt, t1, t2, &o_cur)) : 0;
        if (down_chained && chained != 1) return fail(h, "stage %d: the chain the stride-2 unit prepared for did not run", si + 2);
        if (chained < 0) return 1;
        if (!chained) {
            for (int bi = 1; bi < R; ++bi) {
                snprintf(nm, sizeof nm, "backbone.stage%d.%d", si + 2, bi);
                const std::string P = nm;
                run_unit(h, P, o_cur, B, Ho, Ho, o_nxt, t1, t2);
                float* tmp = o_cur; o_cur = o_nxt; o_nxt = tmp;
            }
        }
        cfeat[si] = o_cur;
        if (h->tap_out[si]) HIPCHK(h, hipMemcpyAsync(h->tap_out[si], o_cur, (size_t)Mo * C * sizeof(float), hipMemcpyDeviceToDevice, h->cur));
        cur = o_cur; curC = C; curH = Ho;
    }
    // neck: models/yolo_nano.py:286-296
    const int W3 = S / 8, W4 = S / 16, W5 = S / 32;
    const long M3 = (long)B * W3 * W3, M4 = (long)B * W4 * W4, M5 = (long)B * W5 * W5;
    TAKE(p3, M3 * NECK); TAKE(p4, M4 * NECK); TAKE(p5, M5 * NECK);
    // the three laterals are independent (p3 is only needed by smooth_1): one grouped launch, or three launches with the large one
    // forked onto a side stream
    bool lat_grouped = false;
    if (h->group_launch && !exact(h)) {
        const Layer* ll[3] = {&L(h, "conv1x1_0"), &L(h, "conv1x1_1"), &L(h, "conv1x1_2")};
        if (ll[0]->ws_hi && ll[1]->ws_hi && ll[2]->ws_hi) {
            // longest K first: a workgroup of the K = 464 lateral runs 15 chunk rounds against 4 for the stride-8 one, and workgroups start
            // in id order - last in the grid, the few long ones were a 25 us tail behind the many short ones
            const Layer* lo[3] = {ll[2], ll[1], ll[0]};
            GemmArgs g3[3] = {pw_args(h, *ll[2], cfeat[2], h->stage_ch[2], 0, M5, p5, NECK, 0, nullptr, 0, 0),
                              pw_args(h, *ll[1], cfeat[1], h->stage_ch[1], 0, M4, p4, NECK, 0, nullptr, 0, 0),
                              pw_args(h, *ll[0], cfeat[0], h->stage_ch[0], 0, M3, p3, NECK, 0, nullptr, 0, 0)};
            lat_grouped = run_pw_group(h, lo, g3, 3, "conv1x1_*");
        }
    }
    if (!lat_grouped) {
        fork_to(h, 0);
        run_pw(h, L(h, "conv1x1_0"), cfeat[0], h->stage_ch[0], 0, M3, p3, NECK, 0, nullptr, 0, 0);
        back_to_main(h);
        run_pw(h, L(h, "conv1x1_1"), cfeat[1], h->stage_ch[1], 0, M4, p4, NECK, 0, nullptr, 0, 0);
        run_pw(h, L(h, "conv1x1_2"), cfeat[2], h->stage_ch[2], 0, M5, p5, NECK, 0, nullptr, 0, 0);
    }
    TAKE(p4a, M4 * NECK); TAKE(p3a, M3 * NECK); TAKE(p4b, M4 * NECK); TAKE(p5a, M5 * NECK);
    run_c3(h, L(h, "smooth_0"), p4, p5, 1, B, W4, W4, p4a);        // p4 + up2(p5)
    join_from(h, 0);
    run_c3(h, L(h, "smooth_1"), p3, p4a, 1, B, W3, W3, p3a);       // p3 + up2(p4)
    // heads: models/yolo_nano.py:299-301.  head k only needs its own pyramid level, so head 1 (the big one) starts on a
    // side stream as soon as smooth_1 is enqueued, head 2 after smooth_2, head 3 stays on the main stream.
    const float* feats[3] = {p3a, p4b, p5a};
    const int Ws[3] = {W3, W4, W5};
    bool fuse_all = fuse_decode && h->fuse_decode && !exact(h);
    for (int hd = 0; hd < 3 && fuse_all; ++hd) {
        snprintf(nm, sizeof nm, "head_det_%d.4", hd + 1);
        fuse_all = head_decode_supported(head_final_args(h, L(h, nm), nullptr, (long)B * Ws[hd] * Ws[hd]), h->grid);
    }
    // small batches: three 16 us fused launches against three 10 us GEMMs + one 7 us decode (bs = 1) - the fusion pays from the
    // traffic it removes, i.e. when the stride-8 head is large (bs >= 4 at 416x416); yn_fuse_decode(2) forces it (tests)
    // ... unless the wider fusion applies (head_tail_group_kernel: layers .2-.4 + decode as ONE launch instead of three: bs = 1 0.602 -> 0.593 ms)
    bool tail_possible = false;
    if (fuse_all && h->tail_fuse && h->dwpw_fuse && h->group_launch && h->grid.C > 32) {
        const GemmArgs gf = head_final_args(h, L(h, "head_det_1.4"), nullptr, (long)B * W3 * W3);
        tail_possible = gf.Npad > 128 && gf.Npad <= 256;
    }
    if (h->fuse_decode_mode != 2 && (long)B * W3 * W3 < 8192 && !tail_possible) fuse_all = false;
    if (fused) *fused = fuse_all;
    auto run_head = [&](int hd) {
        const long M = (long)B * Ws[hd] * Ws[hd];
        float* hA = arena_take(h, (size_t)M * NECK);
        float* hB = arena_take(h, (size_t)M * NECK);
        float* hC = arena_take(h, (size_t)M * NECK);
        if (!hA || !hB || !hC) return 1;
        snprintf(nm, sizeof nm, "head_det_%d", hd + 1);
        const std::string P = nm;
        run_dwpw(h, L(h, P + ".0"), L(h, P + ".1"), feats[hd], NECK, 0, B, Ws[hd], Ws[hd], hA, hB, NECK, 0, nullptr, 0, 0);
        run_dwpw(h, L(h, P + ".2"), L(h, P + ".3"), hB, NECK, 0, B, Ws[hd], Ws[hd], hA, hC, NECK, 0, nullptr, 0, 0);
        const Layer& lf = L(h, P + ".4");
        if (fuse_all) {
            GemmArgs a = head_final_args(h, lf, hC, M);
            set_last_kernel_name("head_decode_kernel");
            Bracket br(h, lf.name + "+decode", 2.0 * M * lf.cin * lf.cout, 4.0 * (M * (double)lf.cin + (double)lf.cin * lf.cout + 6.0 * M * h->grid.A));
            launch_head_decode(a, h->grid, hd, h->cfg.conf_thresh, h->cand_boxes, h->cand_scores, h->cand_cls, h->cur);
        } else {
            run_pw(h, lf, hC, NECK, 0, M, heads[hd], head_ld, 0, nullptr, 0, 0, head_ld);
        }
        return 0;
    };
    // All three heads layer by layer, each layer ONE grouped launch (5 launches instead of 15): the stride-16 / 32 heads ride along with
    // the stride-8 one instead of paying ten launches of 7-20 us for a few microseconds of work.  Needs the split-f16 family.
    bool grouped = h->group_launch && !exact(h);
    const Layer* hl[5][3];
    for (int k = 0; k < 5 && grouped; ++k)
        for (int hd = 0; hd < 3; ++hd) {
            snprintf(nm, sizeof nm, "head_det_%d.%d", hd + 1, k);
            hl[k][hd] = &L(h, nm);
            if ((k == 1 || k == 3 || k == 4) && !hl[k][hd]->ws_hi) grouped = false;
        }
    if (grouped) {
        run_c3(h, L(h, "smooth_2"), p4a, p3a, 2, B, W4, W4, p4b);      // p4 + down(p3)
        run_c3(h, L(h, "smooth_3"), p5, p4b, 2, B, W5, W5, p5a);       // p5 + down(p4)
        float *hA[3], *hB[3], *hC[3];
        for (int hd = 0; hd < 3; ++hd) {
            const size_t M = (size_t)B * Ws[hd] * Ws[hd];
            hA[hd] = arena_take(h, M * NECK); hB[hd] = arena_take(h, M * NECK); hC[hd] = arena_take(h, M * NECK);
            if (!hA[hd] || !hB[hd] || !hC[hd]) return fail(h, "activation arena exhausted (%zu bytes)", h->arena_bytes);
        }
        auto dw_layer = [&](int k, float* const src_own[3], const float* const src_feat[3], float* const dst[3], const char* name) {
            DwArgs d[3];
            for (int hd = 0; hd < 3; ++hd) {
                const Layer& l = *hl[k][hd];
                d[hd] = DwArgs{};
                d[hd].in = src_feat ? src_feat[hd] : src_own[hd]; d[hd].in_ld = NECK; d[hd].in_off = 0; d[hd].w = l.w_packed; d[hd].bias = l.b_packed;
                d[hd].out = dst[hd]; d[hd].out_ld = NECK; d[hd].out_off = 0;
                d[hd].B = B; d[hd].H = Ws[hd]; d[hd].W = Ws[hd]; d[hd].C = l.cout; d[hd].stride = l.stride; d[hd].act = l.act;
            }
            if (!dw_group_ok(d, 3)) return false;
            run_dw_group(h, hl[k], d, 3, name);
            return true;
        };
        auto pw_layer = [&](int k, float* const src[3], float* const dst[3], int dst_ld, int n_store, const char* name) {
            GemmArgs g3[3];
            for (int hd = 0; hd < 3; ++hd)
                g3[hd] = pw_args(h, *hl[k][hd], src[hd], NECK, 0, (long)B * Ws[hd] * Ws[hd], dst[hd], dst_ld, 0, nullptr, 0, 0, n_store);
            return run_pw_group(h, hl[k], g3, 3, name);
        };
        // depthwise + pointwise pairs as one grouped kernel each (the depthwise output never reaches memory), else two grouped launches each
        auto dwpw_layer = [&](int kd, float* const src_own[3], const float* const src_feat[3], float* const dst[3], const char* name) {
            DwPwArgs q[3];
            double fl = 0, by = 0;
            for (int hd = 0; hd < 3; ++hd) {
                const Layer &ld = *hl[kd][hd], &lp = *hl[kd + 1][hd];
                q[hd] = DwPwArgs{};
                q[hd].in = src_feat ? src_feat[hd] : src_own[hd];
                q[hd].wdw = ld.w_packed; q[hd].bdw = ld.b_packed; q[hd].dw_act = ld.act;
                q[hd].Wh = lp.ws_hi; q[hd].Wl = lp.ws_lo; q[hd].bias = lp.b_packed; q[hd].act = lp.act; q[hd].Npad = lp.Npad;
                q[hd].out = dst[hd]; q[hd].B = B; q[hd].H = Ws[hd]; q[hd].W = Ws[hd]; q[hd].C = ld.cout;
                q[hd].ovf = h->range_flags ? h->range_flags + 1 : nullptr;
                if (ld.stride != 1 || ld.cout != NECK || lp.cin != NECK || lp.cout != NECK) return false;
                const double M = (double)B * Ws[hd] * Ws[hd];
                fl += 2.0 * M * NECK * (9.0 + NECK);
                by += 4.0 * (2.0 * M * NECK + (double)NECK * NECK);
            }
            if (!h->dwpw_fuse || !dwpw_group_ok(q, 3)) return false;
            if (dbg_skip(h, name)) return true;
            Bracket br(h, name, fl, by);
            launch_dwpw_group(q, 3, h->cur);
            return true;
        };
        // layers .2 + .3 + .4 + the decode as ONE grouped kernel (head_tail_group_kernel): layer .3's output never reaches memory
        auto tail_layer = [&](float* const src[3]) {
            if (!fuse_all || !h->tail_fuse || !h->dwpw_fuse) return false;
            HeadTailArgs q[3];
            double fl = 0, by = 0;
            for (int hd = 0; hd < 3; ++hd) {
                const Layer &ld = *hl[2][hd], &lp = *hl[3][hd], &lf = *hl[4][hd];
                if (ld.stride != 1 || ld.cout != NECK || lp.cin != NECK || lp.cout != NECK || lp.Npad != NECK || lf.cin != NECK || NECK != 96 || !lf.ws_hi) return false;
                const GemmArgs gf = head_final_args(h, lf, src[hd], (long)B * Ws[hd] * Ws[hd]);
                if (gf.act != 0 || gf.pass) return false;
                q[hd] = HeadTailArgs{};
                q[hd].in = src[hd]; q[hd].wdw = ld.w_packed; q[hd].bdw = ld.b_packed; q[hd].dw_act = ld.act;
                q[hd].Wh = lp.ws_hi; q[hd].Wl = lp.ws_lo; q[hd].bias = lp.b_packed; q[hd].act = lp.act;
                q[hd].Wfh = gf.Wsh; q[hd].Wfl = gf.Wsl; q[hd].fbias = gf.bias; q[hd].Npad = gf.Npad;
                q[hd].B = B; q[hd].H = Ws[hd]; q[hd].W = Ws[hd];
                q[hd].ovf = h->range_flags ? h->range_flags + 1 : nullptr;
                const double M = (double)B * Ws[hd] * Ws[hd];
                fl += 2.0 * M * NECK * (9.0 + NECK) + 2.0 * M * lf.cin * lf.cout;
                by += 4.0 * (M * NECK + (double)NECK * NECK + (double)lf.cin * lf.cout + 6.0 * M * h->grid.A);
            }
            if (!head_tail_ok(q, 3, h->grid)) return false;
            if (dbg_skip(h, "head_det_*.2+3+4")) return true;
            set_last_kernel_name("head_tail_group_kernel");
            Bracket br(h, "head_det_*.2+3+4+decode", fl, by);
            launch_head_tail_group(q, 3, h->grid, h->cfg.conf_thresh, h->cand_boxes, h->cand_scores, h->cand_cls, h->cur);
            return true;
        };
        bool ok = dwpw_layer(0, nullptr, feats, hB, "head_det_*.0+1") || (dw_layer(0, nullptr, feats, hA, "head_det_*.0") && pw_layer(1, hA, hB, NECK, 0, "head_det_*.1"));
        const bool tailed = ok && tail_layer(hB);
        if (ok && !tailed) ok = dwpw_layer(2, hB, nullptr, hC, "head_det_*.2+3") || (dw_layer(2, hB, nullptr, hA, "head_det_*.2") && pw_layer(3, hA, hC, NECK, 0, "head_det_*.3"));
        if (ok && !tailed) {
            if (fuse_all) {
                GemmArgs g3[3];
                double fl = 0, by = 0;
                for (int hd = 0; hd < 3; ++hd) {
                    const long M = (long)B * Ws[hd] * Ws[hd];
                    g3[hd] = head_final_args(h, *hl[4][hd], hC[hd], M);
                    fl += 2.0 * M * hl[4][hd]->cin * hl[4][hd]->cout;
                    by += 4.0 * (M * (double)hl[4][hd]->cin + (double)hl[4][hd]->cin * hl[4][hd]->cout + 6.0 * M * h->grid.A);
                }
                set_last_kernel_name("head_decode_group_kernel");
                Bracket br(h, "head_det_*.4+decode", fl, by);
                if (!dbg_skip(h, "head_det_*.4")) launch_head_decode_group(g3, 3, h->grid, h->cfg.conf_thresh, h->cand_boxes, h->cand_scores, h->cand_cls, h->cur);
            } else {
                float* outs[3] = {heads[0], heads[1], heads[2]};
                ok = pw_layer(4, hC, outs, head_ld, head_ld, "head_det_*.4");
            }
        }
        if (!ok) return fail(h, "grouped head launch: unsupported layer layout");
    } else {
    fork_to(h, 0);
    if (run_head(0)) return fail(h, "activation arena exhausted (%zu bytes)", h->arena_bytes);
    back_to_main(h);
    run_c3(h, L(h, "smooth_2"), p4a, p3a, 2, B, W4, W4, p4b);      // p4 + down(p3)
    fork_to(h, 1);
    if (run_head(1)) return fail(h, "activation arena exhausted (%zu bytes)", h->arena_bytes);
    back_to_main(h);
    run_c3(h, L(h, "smooth_3"), p5, p4b, 2, B, W5, W5, p5a);       // p5 + down(p4)
    if (run_head(2)) return fail(h, "activation arena exhausted (%zu bytes)", h->arena_bytes);
    join_from(h, 0);
    join_from(h, 1);
    }
#undef TAKE
    HIPCHK(h, hipGetLastError());
    return 0;
}

int check_ready(yn_handle* h, int B)
{
    if (!h->folded) return fail(h, "weights not prepared: call yn_fold_bn() after loading parameters");
    if (B <= 0) return fail(h, "batch must be positive (got %d)", B);
    return 0;
}

// run `body` directly, or captured into / replayed from a hipGraph keyed by `key`
template <class F>
int run_maybe_graph(yn_handle* h, const std::vector<uintptr_t>& key, F body)
{
    if (!h->use_graph || h->profiling) return body();
    if (h->stream == nullptr)                               // the legacy default stream cannot be captured; trying leaves the runtime in capture-error state
        return fail(h, "hipGraph capture needs a non-default stream: create the handle on (or yn_set_stream to) a stream of its own");
    for (GraphEntry& g : h->graphs)
        if (g.key == key) { HIPCHK(h, hipGraphLaunch(g.exec, h->stream)); return 0; }
    if (h->autotune) { const int rc0 = body(); if (rc0) return rc0; }      // eager pass: tunes tile configurations, warms up
    hipGraph_t graph = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
    const int rc = body();
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(h, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    GraphEntry ge;
    ge.key = key;
    HIPCHK(h, hipGraphInstantiate(&ge.exec, graph, nullptr, nullptr, 0));
    (void)hipGraphDestroy(graph);
    h->graphs.push_back(ge);
    HIPCHK(h, hipGraphLaunch(ge.exec, h->stream));
    return 0;
}

}  // namespace

// =================================================================================================
#pragma GCC visibility push(default)
extern "C" {

int yn_abi_version(void) { return 2; }       // 2: + yn_range_status, yn_allreduce_grads, yn_tune_save / yn_tune_load, yn_train_get / set_loss_scale

const char* yn_last_error(yn_handle* h) { return h ? h->err.c_str() : g_create_error.c_str(); }

int yn_create(const yn_config* cfg, yn_handle** out)
{
    if (!cfg || !out) return fail(nullptr, "yn_create: null argument");
    *out = nullptr;
    if (cfg->backbone < 0 || cfg->backbone > 3)
        return fail(nullptr, "unsupported backbone id %d (0.5x/1.0x/1.5x/2.0x = 0..3)", cfg->backbone);   // models/yolo_nano.py:35-37
    if (cfg->num_anchors < 1 || cfg->num_anchors > 3) return fail(nullptr, "num_anchors must be 1..3");
    if (cfg->num_classes < 1 || cfg->num_classes > 1024) return fail(nullptr, "num_classes out of range");
    if (cfg->input_size <= 0 || cfg->input_size % 32) return fail(nullptr, "input_size must be a positive multiple of 32");
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) return fail(nullptr, "no HIP device available (%s)", hipGetErrorString(e));
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, "device %d out of range (%d devices)", cfg->device, ndev);
    e = hipSetDevice(cfg->device);
    if (e != hipSuccess) return fail(nullptr, "hipSetDevice(%d): %s", cfg->device, hipGetErrorString(e));
    yn_handle* h = new yn_handle();
    h->cfg = *cfg;
    if (h->cfg.max_batch < 1) h->cfg.max_batch = 1;
    h->stream = (hipStream_t)cfg->stream;
    h->cur = h->stream;
    if (const char* e7 = getenv("YN_EXACT_F32")) h->exact_f32 = atoi(e7) != 0;
    h->nms.prefilter = getenv("YN_NMS_PREFILTER") ? atoi(getenv("YN_NMS_PREFILTER")) : 1;
    h->nms.sweep = getenv("YN_NMS_SWEEP") ? (atoi(getenv("YN_NMS_SWEEP")) != 0) : 1;
    if (const char* e9 = getenv("YN_GROUP")) h->group_launch = atoi(e9) != 0;
    if (const char* e10 = getenv("YN_DOWN_FUSE")) h->down_fuse = atoi(e10) != 0;
    if (const char* e11 = getenv("YN_DWPW_FUSE")) h->dwpw_fuse = atoi(e11) != 0;
    if (const char* e12 = getenv("YN_TAIL_FUSE")) h->tail_fuse = atoi(e12) != 0;
    if (const char* e8 = getenv("YN_FUSE_DECODE")) { h->fuse_decode = atoi(e8) != 0; h->fuse_decode_mode = atoi(e8); }
    if (const char* e6 = getenv("YN_MULTI_STREAM")) h->multi_stream = atoi(e6) != 0;  // A/B switch: fork independent chains onto side streams
    if (const char* e = getenv("YN_CHAIN_PIPE")) h->chain_pipe = atoi(e) < 0 ? 0 : (atoi(e) > 2 ? 2 : atoi(e));
    if (const char* e = getenv("YN_STAGE_FUSE")) h->stage_fuse = atoi(e) < 0 ? 0 : (atoi(e) > 2 ? 2 : atoi(e));
    if (const char* e = getenv("YN_STAGE_PUB")) h->stage_pub_early = atoi(e) != 0;
    if (const char* e = getenv("YN_PW_PIPE")) h->pw_pipe = atoi(e) != 0;
    if (const char* e4 = getenv("YN_UNIT_CHAIN")) h->unit_chain = atoi(e4) < 0 ? 0 : (atoi(e4) > 2 ? 2 : atoi(e4));    // A/B switch for the one-kernel-per-unit chain
    build_layers(h);
    if (set_grid_info(h, cfg->input_size)) { g_create_error = h->err; delete h; return 1; }
    // (hipMemsetAsync on the handle's stream, never hipMemset: ONE operation on the legacy null stream and every later launch of the
    // process on torch's streams serialises against it - measured 38.2 k -> 32.6 k images/s with four handles, single-stream unchanged)
    if (hipMalloc((void**)&h->range_flags, 3 * sizeof(unsigned)) != hipSuccess ||
        hipMemsetAsync(h->range_flags, 0, 3 * sizeof(unsigned), h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) {
        g_create_error = "yn_create: out of device memory"; delete h; return 1;
    }
    if (hipHostMalloc((void**)&h->range_host, sizeof(unsigned), hipHostMallocMapped | hipHostMallocCoherent) != hipSuccess ||      // coherent: the kernel's system-scope store is visible when it is made, not at kernel end (HIP_HOST_COHERENT=0)
        hipHostGetDevicePointer((void**)&h->range_host_dev, h->range_host, 0) != hipSuccess) {
        g_create_error = "yn_create: no pinned host memory for the range flag"; delete h; return 1;
    }
    *h->range_host = 0;
    *out = h;
    return 0;
}

void yn_destroy(yn_handle* h)
{
    if (!h) return;
    DevGuard dev_guard_(h);
    (void)hipStreamSynchronize(h->stream);
    for (auto& kv : h->params) if (kv.second.dev) (void)hipFree(kv.second.dev);
    for (Layer& l : h->layers) {
        if (l.w_packed) (void)hipFree(l.w_packed);
        if (l.b_packed) (void)hipFree(l.b_packed);
        if (l.w_ref) (void)hipFree(l.w_ref);
        if (l.b_ref) (void)hipFree(l.b_ref);
        if (l.ws_hi) (void)hipFree(l.ws_hi);
        if (l.ws_lo) (void)hipFree(l.ws_lo);
    }
    void* ptrs[] = {h->arena, h->cand_boxes, h->cand_scores, h->cand_cls, h->nms.bucket, h->nms.keep, h->nms.sbox,
                    h->nms.seg_count, h->nms.seg_off, h->nms.tile_off, h->nms.large_list, h->nms.matrix, h->heads_int[0], h->loss_partial,
                    h->nms.bucket2, h->nms.sbox2, h->nms.seg_count2, h->nms.tile_off2, h->nms.seg_order, h->nms.ctr, h->nms.seg_sparse, h->nms.work_off, h->nms.pre_sync};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (GraphEntry& g : h->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    for (hipEvent_t e : h->event_pool) (void)hipEventDestroy(e);
    if (h->tune_e0) { (void)hipEventDestroy(h->tune_e0); (void)hipEventDestroy(h->tune_e1); }
    for (hipEvent_t e : h->fj_events) (void)hipEventDestroy(e);
    for (int k = 0; k < 2; ++k) if (h->side[k]) (void)hipStreamDestroy(h->side[k]);
    for (TrainPack& pk : h->tpacks) { if (pk.wp) (void)hipFree(pk.wp); if (pk.bias) (void)hipFree(pk.bias); if (pk.wp_bwd) (void)hipFree(pk.wp_bwd); }
    if (h->range_flags) (void)hipFree(h->range_flags);
    if (h->range_host) (void)hipHostFree(h->range_host);
    if (h->zeros) (void)hipFree(h->zeros);
    for (TrainGraph& g : h->train_graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (h->train_side) (void)hipStreamDestroy(h->train_side);
    for (int i = 0; i < 2; ++i) if (h->train_fork[i]) (void)hipStreamDestroy(h->train_fork[i]);
    for (int i = 0; i < 8; ++i) if (h->fork_ev[i]) (void)hipEventDestroy(h->fork_ev[i]);
    if (h->train_losses) (void)hipFree(h->train_losses);
    if (h->skip_flag) (void)hipFree(h->skip_flag);
    if (h->scale_state) (void)hipFree(h->scale_state);
    if (h->hpack_table) (void)hipFree(h->hpack_table);
    for (HPack& pk : h->hpacks) { void* q[] = {pk.wf, pk.wb, pk.bias, pk.dwf, pk.dwb}; for (void* v : q) if (v) (void)hipFree(v); }
    for (hipEvent_t e : h->train_events) (void)hipEventDestroy(e);
    if (h->train_arena) (void)hipFree(h->train_arena);
    delete h;
}

int yn_set_grid(yn_handle* h, int input_size)
{
    YN_ENTER(h);
    if (set_grid_info(h, input_size)) return 1;
    h->cfg.input_size = input_size;
    return 0;
}

int yn_set_stream(yn_handle* h, void* stream)
{
    YN_ENTER(h);
    if ((hipStream_t)stream != h->stream) {
        HIPCHK(h, hipStreamSynchronize(h->stream));        // the arena is shared: drain the old stream first
        drop_graphs(h);
    }
    h->stream = (hipStream_t)stream;
    h->cur = h->stream;
    return 0;
}

int yn_set_thresholds(yn_handle* h, float conf_thresh, float nms_thresh, int diou_nms)
{
    YN_ENTER(h);
    if (conf_thresh != h->cfg.conf_thresh || nms_thresh != h->cfg.nms_thresh || diou_nms != h->cfg.diou_nms) {
        drop_graphs(h);
    }
    h->cfg.conf_thresh = conf_thresh; h->cfg.nms_thresh = nms_thresh; h->cfg.diou_nms = diou_nms;
    return 0;
}

int yn_num_predictions(yn_handle* h) { return h ? h->grid.N : -1; }

int yn_use_graph(yn_handle* h, int enable) { if (!h) return 1; h->use_graph = enable != 0; return 0; }

int yn_autotune(yn_handle* h, int enable)
{
    YN_ENTER(h);
    h->autotune = enable != 0;                              // off: this handle uses the static heuristic; the shared table stays
    return 0;
}

int yn_set_pw_config(yn_handle* h, int index)
{
    YN_ENTER(h);
    if (index >= pw_config_count()) return fail(h, "yn_set_pw_config: index %d out of range (%d configurations)", index, pw_config_count());
    h->force_pw_cfg = index < 0 ? -1 : index;
    return 0;
}
// The autotuner's table (layer shape -> fastest tile configuration), shared by every handle of the process, to / from a text file:
// one line per entry, the device ordinal left out, so that the ranks of a multi-GPU job can adopt ONE rank's choices (identical
// GPUs, identical shapes: eight ranks timing the same ~40 shapes at once only adds noise to each other's brackets and lets replicas
// end up on different tiles).  Every configuration gives the same bits; this is a speed matter only.
int yn_tune_save(const char* path, int device)
{
    if (!path) return 1;
    FILE* f = fopen(path, "w");
    if (!f) return 1;
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    for (const auto& kv : g_pw_tuned) {
        if (kv.first.empty() || kv.first[0] != device) continue;
        fprintf(f, "%d", kv.second);
        for (size_t i = 1; i < kv.first.size(); ++i) fprintf(f, " %d", kv.first[i]);
        fprintf(f, "\n");
    }
    fclose(f);
    return 0;
}

int yn_tune_load(const char* path, int device)
{
    if (!path) return -1;
    FILE* f = fopen(path, "r");
    if (!f) return -1;
    int n = 0;
    char line[1024];
    std::lock_guard<std::mutex> lk(g_tune_mutex);
    while (fgets(line, sizeof line, f)) {
        std::vector<int> v;
        char* p = line;
        for (;;) {
            char* e = nullptr;
            const long x = strtol(p, &e, 10);
            if (e == p) break;
            v.push_back((int)x);
            p = e;
        }
        if (v.size() < 2 || v[0] < 0 || v[0] >= pw_config_count()) continue;
        std::vector<int> key;
        key.push_back(device);
        key.insert(key.end(), v.begin() + 1, v.end());
        if (!g_pw_tuned.count(key)) { g_pw_tuned[key] = v[0]; ++n; }
    }
    fclose(f);
    return n;
}

int yn_pw_config_count(void) { return pw_config_count(); }
int yn_pw_f32_config_count(void) { return pw_f32_config_count(); }
int yn_nms_prefilter(yn_handle* h, int mode)
{
    YN_ENTER(h);
    if (mode < 0 || mode > 2) return fail(h, "yn_nms_prefilter: mode %d is not 0, 1 or 2", mode);
    if (mode != h->nms.prefilter) drop_graphs(h);
    h->nms.prefilter = mode;
    return 0;
}
int yn_nms_sweep_segments(yn_handle* h, int B, int C)
{
    YN_ENTER(h);
    if (!h->nms.seg_sparse || B <= 0 || C <= 0 || (size_t)B * C > h->nms_seg_cap) return 0;
    std::vector<int32_t> v((size_t)B * C);
    if (hipMemcpyAsync(v.data(), h->nms.seg_sparse, v.size() * sizeof(int32_t), hipMemcpyDeviceToHost, h->stream) != hipSuccess || hipStreamSynchronize(h->stream) != hipSuccess) return -1;
    int k = 0;
    for (int32_t x : v) k += x == 1;
    return k;
}
int yn_nms_sweep(yn_handle* h, int enable)
{
    if (!h) return 1;
    if ((enable != 0) != (h->nms.sweep != 0)) drop_graphs(h);
    h->nms.sweep = enable != 0;
    return 0;
}

int yn_down_fuse(yn_handle* h, int enable)
{
    YN_ENTER(h);
    if ((enable != 0) != h->down_fuse) drop_graphs(h);
    h->down_fuse = enable != 0;
    return 0;
}

int yn_tail_fuse(yn_handle* h, int enable)
{
    YN_ENTER(h);
    if ((enable != 0) != h->tail_fuse) drop_graphs(h);
    h->tail_fuse = enable != 0;
    return 0;
}

int yn_group_launch(yn_handle* h, int enable)
{
    YN_ENTER(h);
    if ((enable != 0) != h->group_launch) drop_graphs(h);
    h->group_launch = enable != 0;
    return 0;
}

int yn_fuse_decode(yn_handle* h, int enable)
{
    YN_ENTER(h);
    if ((enable != 0) != h->fuse_decode || enable != h->fuse_decode_mode) drop_graphs(h);
    h->fuse_decode = enable != 0;
    h->fuse_decode_mode = enable;
    return 0;
}

int yn_exact_f32(yn_handle* h, int enable)
{
    YN_ENTER(h);
    if ((enable != 0) != h->exact_f32) drop_graphs(h);
    h->exact_f32 = enable != 0;
    return 0;
}
int yn_multi_stream(yn_handle* h, int enable) { if (!h) return 1; h->multi_stream = enable != 0; return 0; }
int yn_unit_chain(yn_handle* h, int mode) { if (!h) return 1; h->unit_chain = mode < 0 ? 0 : (mode > 2 ? 2 : mode); return 0; }
int yn_chain_pipe(yn_handle* h, int mode)
{
    if (!h) return 1;
    const int m = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
    if (m != h->chain_pipe) drop_graphs(h);
    h->chain_pipe = m;
    return 0;
}
int yn_stage_fuse(yn_handle* h, int mode, int publish_early)
{
    if (!h) return 1;
    const int m = mode < 0 ? 0 : (mode > 2 ? 2 : mode);
    if (m != h->stage_fuse || (publish_early != 0) != (h->stage_pub_early != 0)) drop_graphs(h);
    h->stage_fuse = m;
    h->stage_pub_early = publish_early != 0;
    return 0;
}
int yn_pw_pipe(yn_handle* h, int enable)
{
    if (!h) return 1;
    if ((enable != 0) != h->pw_pipe) drop_graphs(h);
    h->pw_pipe = enable != 0;
    return 0;
}

int yn_synchronize(yn_handle* h) { YN_ENTER(h); HIPCHK(h, hipStreamSynchronize(h->stream)); return 0; }

// ---- weights -----------------------------------------------------------------------------------
static int load_param_impl(yn_handle* h, const char* key, const void* ptr, const int64_t* shape, int ndim, bool from_dev)
{
    if (!h || !key) return 1;
    if (!ptr) return fail(h, "yn_load_param(%s): null data", key);
    const std::string k = key;
    const bool is_i64 = k.size() > 19 && k.compare(k.size() - 19, 19, "num_batches_tracked") == 0;
    if (is_i64) return 0;                                   // bookkeeping counter, unused in eval
    // validate the key against the architecture
    const size_t dot = k.rfind('.');
    if (dot == std::string::npos) return fail(h, "unknown state-dict key '%s'", key);
    const std::string prefix = k.substr(0, dot), leaf = k.substr(dot + 1);
    const Layer* owner = nullptr;
    bool is_bn = false;
    auto it = h->by_conv.find(prefix);
    if (it != h->by_conv.end() && (leaf == "weight" || leaf == "bias")) owner = &h->layers[it->second];
    if (!owner) {
        for (const Layer& l : h->layers)
            if (!l.bn.empty() && l.bn == prefix) { owner = &l; is_bn = true; break; }
    }
    if (!owner) return fail(h, "unexpected state-dict key '%s'", key);
    size_t numel = 1;
    for (int i = 0; i < ndim; ++i) numel *= (size_t)shape[i];
    size_t expect;
    if (is_bn || leaf == "bias") expect = owner->cout;
    else if (owner->kind == K_DW) expect = (size_t)owner->cout * 9;
    else if (owner->kind == K_PW) expect = (size_t)owner->cout * owner->cin;
    else expect = (size_t)owner->cout * owner->cin * 9;
    if (numel != expect) return fail(h, "size mismatch for '%s': got %zu elements, expected %zu", key, numel, expect);
    Param& p = h->params[k];
    const size_t bytes = numel * sizeof(float);
    if (p.bytes != bytes) {
        if (p.dev) { HIPCHK(h, hipStreamSynchronize(h->stream)); HIPCHK(h, hipFree(p.dev)); p.dev = nullptr; }
        HIPCHK(h, hipMalloc(&p.dev, bytes));
        p.bytes = bytes;
    }
    p.numel = numel;
    p.shape.assign(shape, shape + ndim);
    HIPCHK(h, hipMemcpyAsync(p.dev, ptr, bytes, from_dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, h->stream));
    if (!from_dev) HIPCHK(h, hipStreamSynchronize(h->stream));   // host buffer may be freed by the caller
    h->folded = false;
    return 0;
}

int yn_load_param(yn_handle* h, const char* key, const void* host_ptr, const int64_t* shape, int ndim)
{
    return load_param_impl(h, key, host_ptr, shape, ndim, false);
}

int yn_load_param_dev(yn_handle* h, const char* key, const void* dev_ptr, const int64_t* shape, int ndim)
{
    return load_param_impl(h, key, dev_ptr, shape, ndim, true);
}

int yn_fold_bn(yn_handle* h)
{
    YN_ENTER(h);
    if (h->tP)                                              // after training: the flat buffer holds the current parameters
        for (const auto& kv : h->toff) {
            const Param* p = find_param(h, kv.first);
            if (p) HIPCHK(h, hipMemcpyAsync(p->dev, h->tP + kv.second, p->numel * sizeof(float), hipMemcpyDeviceToDevice, h->stream));
        }
    HIPCHK(h, hipMemsetAsync(h->range_flags, 0, sizeof(unsigned), h->stream));
    for (Layer& l : h->layers) {
        const Param* w = find_param(h, l.conv + ".weight");
        if (!w) return fail(h, "missing parameter '%s.weight'", l.conv.c_str());
        const Param* b = find_param(h, l.conv + ".bias");
        const Param *g = nullptr, *be = nullptr, *mu = nullptr, *var = nullptr;
        if (!l.bn.empty()) {
            g = find_param(h, l.bn + ".weight"); be = find_param(h, l.bn + ".bias");
            mu = find_param(h, l.bn + ".running_mean"); var = find_param(h, l.bn + ".running_var");
            const int have = (g != nullptr) + (be != nullptr) + (mu != nullptr) + (var != nullptr);
            if (have != 0 && have != 4) return fail(h, "incomplete BatchNorm parameters for '%s'", l.bn.c_str());
            if (have == 0 && !b) return fail(h, "'%s' has neither BatchNorm statistics nor a folded bias", l.conv.c_str());
        } else if (!b) return fail(h, "missing parameter '%s.bias'", l.conv.c_str());
        FoldArgs a{};
        a.w = (const float*)w->dev; a.b = b ? (const float*)b->dev : nullptr;
        a.gamma = g ? (const float*)g->dev : nullptr; a.beta = be ? (const float*)be->dev : nullptr;
        a.mean = mu ? (const float*)mu->dev : nullptr; a.var = var ? (const float*)var->dev : nullptr;
        a.eps = 1e-5f;
        a.Cout = l.cout; a.Cin = l.cin;
        size_t packed_floats;
        if (l.kind == K_DW) { a.kind = 1; a.kk = 9; packed_floats = (size_t)9 * l.cout; l.Npad = l.cout; l.Kp = 9; }
        else if (l.kind == K_STEM) { a.kind = 2; a.kk = 9; packed_floats = (size_t)27 * l.cout; l.Npad = l.cout; l.Kp = 27; }
        else {
            a.kind = 0; a.kk = (l.kind == K_DENSE3) ? 9 : 1;
            const int K = l.cin * a.kk;
            l.Kp = (K + 1) & ~1; l.Npad = (l.cout + 31) & ~31;
            packed_floats = (size_t)l.Kp * l.Npad;
        }
        a.Kp = l.Kp; a.Npad = l.Npad;
        if (!l.w_packed) {
            HIPCHK(h, hipMalloc((void**)&l.w_packed, packed_floats * sizeof(float)));
            HIPCHK(h, hipMalloc((void**)&l.b_packed, (size_t)((l.Npad + 31) & ~31) * sizeof(float)));
            HIPCHK(h, hipMalloc((void**)&l.w_ref, w->numel * sizeof(float)));
            HIPCHK(h, hipMalloc((void**)&l.b_ref, (size_t)l.cout * sizeof(float)));
            l.w_numel = w->numel;
        }
        if ((l.kind == K_DENSE3 || l.kind == K_PW) && !l.ws_hi) {
            l.ws_bytes = (size_t)(l.kind == K_DENSE3 ? 9 : 1) * ((l.cin + 7) / 8) * l.Npad * 8 * sizeof(_Float16);
            HIPCHK(h, hipMalloc(&l.ws_hi, l.ws_bytes));
            HIPCHK(h, hipMalloc(&l.ws_lo, l.ws_bytes));
        }
        if (l.ws_hi) {
            HIPCHK(h, hipMemsetAsync(l.ws_hi, 0, l.ws_bytes, h->stream));
            HIPCHK(h, hipMemsetAsync(l.ws_lo, 0, l.ws_bytes, h->stream));
        }
        a.ws_hi = l.ws_hi; a.ws_lo = l.ws_lo; a.w_ovf = h->range_flags;
        HIPCHK(h, hipMemsetAsync(l.w_packed, 0, packed_floats * sizeof(float), h->stream));
        HIPCHK(h, hipMemsetAsync(l.b_packed, 0, (size_t)((l.Npad + 31) & ~31) * sizeof(float), h->stream));
        a.w_ref = l.w_ref; a.b_ref = l.b_ref; a.w_packed = l.w_packed; a.b_packed = l.b_packed;
        launch_fold_pack(a, h->stream);
    }
    HIPCHK(h, hipGetLastError());
    // Range guard of the split-f16 family (yn_device.h): a folded weight w * gamma / sqrt(var + eps) that does not fit x = hi + lo * 2^-11
    // (|w| >= 65504) would make its hi part inf and every product NaN where the reference's fp32 conv is finite.  One flag, read once per
    // fold: the handle then runs the f32-MFMA family as under yn_exact_f32(1) (yn_range_status reports it).
    unsigned wflag = 0;
    HIPCHK(h, hipMemcpyAsync(&wflag, h->range_flags, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if ((wflag != 0) != h->range_fallback) drop_graphs(h);
    h->range_fallback = wflag != 0;
    h->folded = true;
    return 0;
}

// The out-of-band form of yn_infer's range mark: compact_kernel set the handle's pinned word, nobody has acknowledged it through
// yn_range_status yet.  No synchronisation: the word is host memory.
static int range_pending(yn_handle* h, const char* who)
{
    // (not under yn_exact_f32: the f32-MFMA family cannot raise the flag, and running again under it IS the documented recovery - ADVICE r5)
    if (!exact(h) && h->range_host && __atomic_load_n(h->range_host, __ATOMIC_RELAXED)) {
        fail(h, "%s: an earlier yn_infer split an activation >= 65504 (split-f16 range): its results are invalid - call yn_range_status to acknowledge, "
                "then yn_exact_f32(h, 1) and run again", who);
        return YN_STATUS_RANGE;
    }
    return 0;
}

int yn_range_status(yn_handle* h, int* weights_exceed_f16, int* activation_overflow)
{
    YN_ENTER(h);
    unsigned f = 0;
    HIPCHK(h, hipMemcpyAsync(&f, h->range_flags + 1, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (f) HIPCHK(h, hipMemsetAsync(h->range_flags + 1, 0, sizeof(unsigned), h->stream));
    if (h->range_host) __atomic_store_n(h->range_host, 0u, __ATOMIC_RELAXED);      // acknowledged (the stream is idle: no compact_kernel can still set it)
    if (h->stage_sync) {                                                            // stage_pipe_kernel's bounded waits (kernels_stage.hip): an expired one marks the launch
        unsigned tmo = 0;
        HIPCHK(h, hipMemcpyAsync(&tmo, h->stage_sync + STAGE_TIMEOUT, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
        if (tmo) {
            HIPCHK(h, hipMemsetAsync(h->stage_sync + STAGE_TIMEOUT, 0, sizeof(unsigned), h->stream));
            return fail(h, "stage_pipe_kernel: a bounded wait for a tile's ready flag expired (2 s): the results of that call are invalid; yn_stage_fuse(h, 0, 1) runs the units one launch each");
        }
    }
    if (weights_exceed_f16) *weights_exceed_f16 = h->range_fallback ? 1 : 0;
    if (activation_overflow) *activation_overflow = f ? 1 : 0;
    return 0;
}

int yn_get_folded(yn_handle* h, const char* conv_key, float* host_weight, float* host_bias)
{
    YN_ENTER(h);
    if (!h->folded) return fail(h, "yn_get_folded before yn_fold_bn");
    auto it = h->by_conv.find(conv_key);
    if (it == h->by_conv.end()) return fail(h, "unknown conv '%s'", conv_key);
    const Layer& l = h->layers[it->second];
    HIPCHK(h, hipMemcpyAsync(host_weight, l.w_ref, l.w_numel * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipMemcpyAsync(host_bias, l.b_ref, (size_t)l.cout * sizeof(float), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return 0;
}

// ---- network -----------------------------------------------------------------------------------
int yn_forward_raw(yn_handle* h, const float* x_dev, int B, float* head_s8, float* head_s16, float* head_s32)
{
    YN_ENTER(h);
    if (check_ready(h, B)) return 1;
    if (ensure_arena(h, B, h->grid.S)) return 1;
    float* const heads[3] = {head_s8, head_s16, head_s32};
    std::vector<uintptr_t> key = {1, (uintptr_t)B, (uintptr_t)h->grid.S, (uintptr_t)x_dev, (uintptr_t)head_s8, (uintptr_t)head_s16, (uintptr_t)head_s32};
    return run_maybe_graph(h, key, [&]() { return run_network(h, x_dev, B, heads, h->head_ch); });
}

int yn_forward_taps(yn_handle* h, const float* x_dev, int B, float* c3_dev, float* c4_dev, float* c5_dev)
{
    YN_ENTER(h);
    if (check_ready(h, B)) return 1;
    if (!c3_dev || !c4_dev || !c5_dev) return fail(h, "yn_forward_taps: null output");
    if (ensure_arena(h, B, h->grid.S) || ensure_heads(h, B)) return 1;
    float* const heads[3] = {h->heads_int[0], h->heads_int[1], h->heads_int[2]};
    h->tap_out[0] = c3_dev; h->tap_out[1] = c4_dev; h->tap_out[2] = c5_dev;
    const int rc = run_network(h, x_dev, B, heads, (h->head_ch + 3) & ~3);       // eager: the taps are not part of any captured graph
    h->tap_out[0] = h->tap_out[1] = h->tap_out[2] = nullptr;
    return rc;
}

int yn_score_full(yn_handle* h, const float* h8, const float* h16, const float* h32, int B, float* all_bbox, float* all_class)
{
    YN_ENTER(h);
    const float* const heads[3] = {h8, h16, h32};
    launch_score_full(heads, h->grid, B, all_bbox, all_class, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_decode_boxes(yn_handle* h, const float* txtytwth, int B, float* xyxy)
{
    YN_ENTER(h);
    launch_decode_boxes(txtytwth, h->grid, B, xyxy, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_create_grid(yn_handle* h, int input_size, float* grid, float* stride, float* anchor)
{
    YN_ENTER(h);
    if (input_size <= 0 || input_size % 32) return fail(h, "input_size must be a positive multiple of 32");
    const int A = h->cfg.num_anchors;
    size_t cell = 0;
    for (int s = 0; s < 3; ++s) {
        const int st = 8 << s, w = input_size / st;
        for (int y = 0; y < w; ++y)
            for (int x = 0; x < w; ++x, ++cell) {
                grid[cell * 2 + 0] = (float)x; grid[cell * 2 + 1] = (float)y;      // (x, y): models/yolo_nano.py:96
                for (int a = 0; a < A; ++a) {
                    stride[(cell * A + a) * 2 + 0] = (float)st; stride[(cell * A + a) * 2 + 1] = (float)st;
                    anchor[(cell * A + a) * 2 + 0] = h->cfg.anchors[(s * A + a) * 2 + 0];
                    anchor[(cell * A + a) * 2 + 1] = h->cfg.anchors[(s * A + a) * 2 + 1];
                }
            }
    }
    return 0;
}

// ---- post-processing ---------------------------------------------------------------------------
int yn_nms(yn_handle* h, const float* dets, const float* scores, int n, float nms_thresh, int diou, int32_t* keep, int32_t* count)
{
    YN_ENTER(h);
    if (n < 0) return fail(h, "yn_nms: negative n");
    if (ensure_post(h, 1, n > 0 ? n : 1, 1)) return 1;
    launch_nms_single(dets, scores, n, nms_thresh, diou, h->nms.bucket, h->nms.sbox, h->nms.matrix, keep, count, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_postprocess(yn_handle* h, const float* all_local, const float* all_conf, int B, int N, int C,
                   float* out_boxes, float* out_scores, int32_t* out_cls, int32_t* out_index, int32_t* count)
{
    YN_ENTER(h);
    if (B <= 0 || N < 0 || C <= 0) return fail(h, "yn_postprocess: bad sizes B=%d N=%d C=%d", B, N, C);
    if (N == 0) { HIPCHK(h, hipMemsetAsync(count, 0, sizeof(int32_t) * B, h->stream)); return 0; }
    if (ensure_post(h, B, N, C)) return 1;
    launch_argmax_cand(all_local, all_conf, B, N, C, h->cfg.conf_thresh, h->cand_boxes, h->cand_scores, h->cand_cls, h->stream);
    launch_nms_pipeline(h->cand_boxes, h->cand_scores, h->cand_cls, B, N, C, h->cfg.nms_thresh, h->cfg.diou_nms, h->nms,
                        out_boxes, out_scores, out_cls, out_index, count, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_preprocess(yn_handle* h, const uint8_t* img, int h0, int w0, int rw, int rh, int left, int top, int side,
                  const float* mean, const float* stdv, float* x)
{
    YN_ENTER(h);
    if (!img || !x || !mean || !stdv) return fail(h, "yn_preprocess: null pointer");
    if (h0 <= 0 || w0 <= 0 || rw <= 0 || rh <= 0 || side <= 0 || left < 0 || top < 0 || left + rw > side || top + rh > side)
        return fail(h, "yn_preprocess: bad geometry (%dx%d -> %dx%d at (%d,%d) in %d)", w0, h0, rw, rh, left, top, side);
    for (int c = 0; c < 3; ++c)
        if (!(stdv[c] > 0.0f)) return fail(h, "yn_preprocess: std must be positive");
    launch_preprocess(img, h0, w0, rw, rh, left, top, side, mean, stdv, x, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_preprocess_batch(yn_handle* h, int n, const uint8_t* const* imgs, const int32_t* geom, int side, const float* mean, const float* stdv, float* x)
{
    YN_ENTER(h);
    if (n == 0) return 0;                                   // an empty batch is not an error
    if (n < 0 || !imgs || !geom || !x || !mean || !stdv || side <= 0) return fail(h, "yn_preprocess_batch: bad arguments");
    for (int c = 0; c < 3; ++c)
        if (!(stdv[c] > 0.0f)) return fail(h, "yn_preprocess_batch: std must be positive");
    for (int i = 0; i < n; ++i) {
        const int32_t* g = geom + (size_t)i * 6;
        if (!imgs[i] || g[0] <= 0 || g[1] <= 0 || g[2] <= 0 || g[3] <= 0 || g[4] < 0 || g[5] < 0 || g[4] + g[2] > side || g[5] + g[3] > side)
            return fail(h, "yn_preprocess_batch: bad geometry for image %d", i);
    }
    launch_preprocess_batch(n, imgs, geom, side, mean, stdv, x, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_nms_merge(yn_handle* h, const float* boxes, const float* scores, const int32_t* cls, int n, int num_classes, float nms_thresh, int diou,
                 float* out_boxes, float* out_scores, int32_t* out_cls, int32_t* out_index, int32_t* count)
{
    YN_ENTER(h);
    if (n < 0 || num_classes <= 0 || !count) return fail(h, "yn_nms_merge: bad arguments");
    if (n == 0) { HIPCHK(h, hipMemsetAsync(count, 0, sizeof(int32_t), h->stream)); return 0; }
    if (ensure_post(h, 1, n, num_classes)) return 1;
    launch_nms_pipeline(boxes, scores, cls, 1, n, num_classes, nms_thresh, diou, h->nms, out_boxes, out_scores, out_cls, out_index, count, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_infer(yn_handle* h, const float* x_dev, int B, float* out_boxes, float* out_scores, int32_t* out_cls,
             int32_t* out_index, int32_t* count)
{
    YN_ENTER(h);
    if (check_ready(h, B)) return 1;
    if (int rp = range_pending(h, "yn_infer")) return rp;
    GridInfo g = h->grid;
    g.head_ld = (h->head_ch + 3) & ~3;
    if (ensure_arena(h, B, g.S) || ensure_post(h, B, g.N, g.C) || ensure_heads(h, B)) return 1;
    std::vector<uintptr_t> key = {2, (uintptr_t)B, (uintptr_t)g.S, (uintptr_t)x_dev, (uintptr_t)out_boxes, (uintptr_t)out_scores,
                                  (uintptr_t)out_cls, (uintptr_t)out_index, (uintptr_t)count};
    return run_maybe_graph(h, key, [&]() {
        float* const heads[3] = {h->heads_int[0], h->heads_int[1], h->heads_int[2]};
        bool fused = false;
        if (run_network(h, x_dev, B, heads, g.head_ld, true, &fused)) return 1;
        const float* const ch[3] = {heads[0], heads[1], heads[2]};
        if (!fused) {
            set_last_kernel_name("decode_kernel<false>");
            Bracket br(h, "decode", 0.0, 4.0 * B * ((double)(g.N / g.A) * h->head_ch + 6.0 * g.N));
            launch_decode_cand(ch, g, B, h->cfg.conf_thresh, h->cand_boxes, h->cand_scores, h->cand_cls, h->stream);
        }
        {
            // one profiling bracket per NMS kernel (the hook closes the previous one); bytes: candidate arrays read / written once
            struct Ctx { yn_handle* h; Bracket* cur; double bytes; } ctx{h, nullptr, 4.0 * B * 12.0 * g.N / 5.0};
            NmsHook hook{[](void* c, const char* k) {
                Ctx* x = (Ctx*)c;
                delete x->cur;
                set_last_kernel_name(k);
                x->cur = new Bracket(x->h, std::string("nms.") + k, 0.0, x->bytes);
            }, &ctx};
            NmsWork wk = h->nms;
            wk.ovf = exact(h) ? nullptr : h->range_flags + 1;         // the network's range flag rides out with the counts (compact_kernel)
            wk.ovf_host = exact(h) ? nullptr : h->range_host_dev;     // ... and into the handle's pinned word (range_pending)
            launch_nms_pipeline(h->cand_boxes, h->cand_scores, h->cand_cls, B, g.N, g.C, h->cfg.nms_thresh, h->cfg.diou_nms, wk,
                                out_boxes, out_scores, out_cls, out_index, count, h->stream, h->profiling ? &hook : nullptr);
            delete ctx.cur;
        }
        HIPCHK(h, hipGetLastError());
        return 0;
    });
}

int yn_pack_detections(yn_handle* h, const float* out_boxes, const float* out_scores, const int32_t* out_cls, const int32_t* count,
                       int B, int N, float* rec_dev, int32_t* offsets_dev)
{
    YN_ENTER(h);
    if (B <= 0 || N <= 0 || !out_boxes || !out_scores || !out_cls || !count || !rec_dev || !offsets_dev)
        return fail(h, "yn_pack_detections: bad arguments");
    if (int rp = range_pending(h, "yn_pack_detections")) return rp;
    launch_pack(out_boxes, out_scores, out_cls, count, B, N, rec_dev, offsets_dev, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// ---- training loss ----------------------------------------------------------
static int ensure_loss(yn_handle* h, int B)
{
    const size_t need = (size_t)loss_num_blocks(h->grid, B) * 4;
    if (need <= h->loss_partial_cap) return 0;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    drop_train_graphs(h);                                   // a captured fp16 step carries loss_partial's address (launch_loss_h16)
    if (h->loss_partial) HIPCHK(h, hipFree(h->loss_partial));
    HIPCHK(h, hipMalloc((void**)&h->loss_partial, need * sizeof(float)));
    h->loss_partial_cap = need;
    return 0;
}

int yn_loss(yn_handle* h, const float* conf, const float* cls, const float* txtytwth, const float* target, int B,
            float* losses, float* g_conf, float* g_cls, float* g_txtytwth)
{
    YN_ENTER(h);
    if (B <= 0) return fail(h, "yn_loss: batch must be positive");
    if ((g_conf != nullptr) != (g_cls != nullptr) || (g_conf != nullptr) != (g_txtytwth != nullptr))
        return fail(h, "yn_loss: pass all three gradient buffers or none");
    if (ensure_loss(h, B)) return 1;
    if (g_cls) HIPCHK(h, hipMemsetAsync(g_cls, 0, (size_t)B * h->grid.N * h->grid.C * sizeof(float), h->stream));   // the kernel only writes the positives' class gradients
    launch_loss(conf, cls, txtytwth, nullptr, nullptr, target, h->grid, B, h->loss_partial, losses, g_conf, g_cls, g_txtytwth, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_loss_heads(yn_handle* h, const float* head_s8, const float* head_s16, const float* head_s32, const float* target, int B,
                  float* losses, float* g_s8, float* g_s16, float* g_s32)
{
    YN_ENTER(h);
    if (B <= 0) return fail(h, "yn_loss_heads: batch must be positive");
    if ((g_s8 != nullptr) != (g_s16 != nullptr) || (g_s8 != nullptr) != (g_s32 != nullptr))
        return fail(h, "yn_loss_heads: pass all three gradient buffers or none");
    if (ensure_loss(h, B)) return 1;
    const float* const head[3] = {head_s8, head_s16, head_s32};
    float* const ghead[3] = {g_s8, g_s16, g_s32};
    if (g_s8)                                                   // the kernel only writes the positives' class gradients
        for (int k = 0; k < 3; ++k) HIPCHK(h, hipMemsetAsync(ghead[k], 0, (size_t)B * h->grid.hw[k] * h->grid.head_ld * sizeof(float), h->stream));
    launch_loss(nullptr, nullptr, nullptr, head, ghead, target, h->grid, B, h->loss_partial, losses, nullptr, nullptr, nullptr, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_make_targets(yn_handle* h, const double* labels_dev, const int32_t* offsets_dev, int B, const double* anchors_host, float* target_dev)
{
    YN_ENTER(h);
    if (B <= 0 || !offsets_dev || !target_dev || !anchors_host) return fail(h, "yn_make_targets: bad arguments");
    if (h->grid.A != 3) return fail(h, "yn_make_targets: the label assigner is defined for 3 anchors per scale (got %d)", h->grid.A);
    HIPCHK(h, hipMemsetAsync(target_dev, 0, (size_t)B * h->grid.N * 11 * sizeof(float), h->stream));
    launch_make_targets(labels_dev, offsets_dev, B, anchors_host, h->grid, target_dev, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_ema_update(yn_handle* h, float* ema_dev, const float* model_dev, int64_t n, double decay)
{
    YN_ENTER(h);
    if (n < 0 || (n > 0 && (!ema_dev || !model_dev))) return fail(h, "yn_ema_update: bad arguments");
    if (n == 0) return 0;
    launch_ema(ema_dev, model_dev, (long)n, (float)decay, (float)(1.0 - decay), h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_sgd_step(yn_handle* h, float* params, const float* grads, float* momentum_buf, int64_t n,
                float lr, float momentum, float weight_decay, float grad_scale, int first_step)
{
    YN_ENTER(h);
    if (n < 0 || !params || !grads || !momentum_buf) return fail(h, "yn_sgd_step: bad arguments");
    if (((uintptr_t)params | (uintptr_t)grads | (uintptr_t)momentum_buf) & 15) return fail(h, "yn_sgd_step: buffers must be 16-byte aligned");
    if (!h->skip_flag) {
        HIPCHK(h, hipMalloc((void**)&h->skip_flag, 2 * sizeof(int)));
        HIPCHK(h, hipMemsetAsync(h->skip_flag, 0, 2 * sizeof(int), h->stream));
    }
    launch_sgd(params, grads, momentum_buf, (long)n, lr, momentum, weight_decay, grad_scale, first_step, h->skip_flag, h->stream);
    // the fp16 step's loss scale follows the finite-scan of THIS bucket (after a data-parallel all-reduce: the same decision on every rank)
    if (h->scale_state && grads == h->tG) launch_hscale_update(h->scale_state, h->skip_flag, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_train_get_loss_scale(yn_handle* h, float* scale, float* clean_steps)
{
    YN_ENTER(h);
    float v[3] = {h->loss_scale_init >= 1.0f ? h->loss_scale_init : 1024.0f, 0.0f, h->loss_scale_clean};
    if (h->scale_state) {
        HIPCHK(h, hipMemcpyAsync(v, h->scale_state, sizeof v, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    if (scale) *scale = v[0];
    if (clean_steps) *clean_steps = v[2];
    return 0;
}

int yn_train_set_loss_scale(yn_handle* h, float scale, float clean_steps)
{
    YN_ENTER(h);
    // hscale_update_kernel halves down to 1 and doubles up to 65536; a larger restored value (up to 2^30: the overflow tests start there)
    // is accepted and can only shrink from then on
    if (!(scale >= 1.0f) || !(scale <= 1073741824.0f) || !(clean_steps >= 0.0f)) return fail(h, "yn_train_set_loss_scale: scale must lie in [1, 2^30], clean_steps >= 0");
    h->loss_scale_init = scale; h->loss_scale_clean = clean_steps;
    if (h->scale_state) {
        // all five words: a restored scale must not inherit the overflow flag / pending mark of an unsettled do_update = 0 step
        const float v[5] = {scale, 1.0f / scale, clean_steps, 0.0f, 0.0f};
        HIPCHK(h, hipMemcpyAsync(h->scale_state, v, sizeof v, hipMemcpyHostToDevice, h->stream));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return 0;
}

// ---- single operators ----------------------------------------------------------------------------
namespace {
struct TmpLayer {
    Layer l;
    yn_handle* h;
    int rc = 0;
    TmpLayer(yn_handle* h_, int kind, int cin, int cout, int stride, int act, const float* w, const float* bias) : h(h_)
    {
        h->cur = h->stream;
        l.name = "op"; l.kind = kind; l.cin = cin; l.cout = cout; l.stride = stride; l.act = act;
        FoldArgs a{};
        a.w = w; a.b = bias; a.eps = 1e-5f; a.Cout = cout; a.Cin = cin;
        size_t packed;
        if (kind == K_DW) { a.kind = 1; a.kk = 9; packed = (size_t)9 * cout; l.Npad = cout; l.Kp = 9; }
        else if (kind == K_STEM) { a.kind = 2; a.kk = 9; packed = (size_t)27 * cout; l.Npad = cout; l.Kp = 27; }
        else { a.kind = 0; a.kk = kind == K_DENSE3 ? 9 : 1; const int K = cin * a.kk; l.Kp = (K + 1) & ~1; l.Npad = (cout + 31) & ~31; packed = (size_t)l.Kp * l.Npad; }
        a.Kp = l.Kp; a.Npad = l.Npad;
        const size_t bfl = (size_t)((l.Npad + 31) & ~31);
        if (hipMalloc((void**)&l.w_packed, packed * sizeof(float)) != hipSuccess || hipMalloc((void**)&l.b_packed, bfl * sizeof(float)) != hipSuccess) { rc = 1; return; }
        (void)hipMemsetAsync(l.w_packed, 0, packed * sizeof(float), h->stream);
        (void)hipMemsetAsync(l.b_packed, 0, bfl * sizeof(float), h->stream);
        a.w_packed = l.w_packed; a.b_packed = l.b_packed;
        if (kind == K_DENSE3 || kind == K_PW) {             // the split-f16 packs the network's GEMM-shaped layers run on
            l.ws_bytes = (size_t)(kind == K_DENSE3 ? 9 : 1) * ((cin + 7) / 8) * l.Npad * 8 * sizeof(_Float16);
            if (hipMalloc(&l.ws_hi, l.ws_bytes) != hipSuccess || hipMalloc(&l.ws_lo, l.ws_bytes) != hipSuccess) { rc = 1; return; }
            (void)hipMemsetAsync(l.ws_hi, 0, l.ws_bytes, h->stream);
            (void)hipMemsetAsync(l.ws_lo, 0, l.ws_bytes, h->stream);
            a.ws_hi = l.ws_hi; a.ws_lo = l.ws_lo; a.w_ovf = h->range_flags + 2;
            (void)hipMemsetAsync(h->range_flags + 2, 0, sizeof(unsigned), h->stream);
        }
        launch_fold_pack(a, h->stream);
        if (l.ws_hi) {                                      // a weight outside the split's range: this operator runs on the f32-MFMA family
            unsigned wflag = 0;
            (void)hipMemcpyAsync(&wflag, h->range_flags + 2, sizeof(unsigned), hipMemcpyDeviceToHost, h->stream);
            if (hipStreamSynchronize(h->stream) != hipSuccess) { rc = 1; return; }
            if (wflag) { (void)hipFree(l.ws_hi); (void)hipFree(l.ws_lo); l.ws_hi = l.ws_lo = nullptr; }
        }
    }
    ~TmpLayer()
    {
        (void)hipStreamSynchronize(h->stream);
        if (l.w_packed) (void)hipFree(l.w_packed);
        if (l.b_packed) (void)hipFree(l.b_packed);
        if (l.ws_hi) (void)hipFree(l.ws_hi);
        if (l.ws_lo) (void)hipFree(l.ws_lo);
    }
};
}  // namespace

int yn_op_dwconv3x3(yn_handle* h, const float* x, int B, int H, int W, int C, int stride, const float* w, const float* bias, int act, float* y)
{
    YN_ENTER(h);
    if (C & 1) return fail(h, "yn_op_dwconv3x3: C must be even");
    TmpLayer t(h, K_DW, C, C, stride, act, w, bias);
    if (t.rc) return fail(h, "yn_op_dwconv3x3: out of memory");
    run_dw(h, t.l, x, C, 0, B, H, W, y, C, 0);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_op_pwconv(yn_handle* h, const float* x, int B, int H, int W, int Cin, int Cout, const float* w, const float* bias, int act, float* y)
{
    YN_ENTER(h);
    if (Cin & 1) return fail(h, "yn_op_pwconv: Cin must be even");
    TmpLayer t(h, K_PW, Cin, Cout, 1, act, w, bias);
    if (t.rc) return fail(h, "yn_op_pwconv: out of memory");
    run_pw(h, t.l, x, Cin, 0, (long)B * H * W, y, Cout, 0, nullptr, 0, 0);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_op_pwconv_shuffle(yn_handle* h, const float* x, const float* pass, int B, int H, int W, int Cin, int Cout,
                         const float* w, const float* bias, int act, float* y)
{
    YN_ENTER(h);
    if ((Cin & 1) || (Cout & 1)) return fail(h, "yn_op_pwconv_shuffle: Cin and Cout must be even");
    if (!pass) return fail(h, "yn_op_pwconv_shuffle: null pass-through tensor");
    TmpLayer t(h, K_PW, Cin, Cout, 1, act, w, bias);
    if (t.rc) return fail(h, "yn_op_pwconv_shuffle: out of memory");
    run_pw(h, t.l, x, Cin, 0, (long)B * H * W, y, 2 * Cout, 0, pass, Cout, 0);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_op_conv3x3(yn_handle* h, const float* x, const float* x2, int resample, int B, int H, int W, int Cin, int Cout,
                  const float* w, const float* bias, int act, float* y)
{
    YN_ENTER(h);
    if (Cin % 32) return fail(h, "yn_op_conv3x3: Cin must be a multiple of 32");
    if (resample && !x2) return fail(h, "yn_op_conv3x3: resample without x2");
    TmpLayer t(h, K_DENSE3, Cin, Cout, 1, act, w, bias);
    if (t.rc) return fail(h, "yn_op_conv3x3: out of memory");
    run_c3(h, t.l, x, x2, resample, B, H, W, y);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_op_stem(yn_handle* h, const float* x, int B, int H, int W, int Cout, const float* w, const float* bias, int act, float* y)
{
    YN_ENTER(h);
    if (Cout != 24) return fail(h, "yn_op_stem: Cout must be 24");
    TmpLayer t(h, K_STEM, 3, Cout, 2, act, w, bias);
    if (t.rc) return fail(h, "yn_op_stem: out of memory");
    launch_stem(x, B, H, W, t.l.w_packed, t.l.b_packed, Cout, act, y, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_op_maxpool3x3s2(yn_handle* h, const float* x, int B, int H, int W, int C, float* y)
{
    YN_ENTER(h);
    if (C % 4) return fail(h, "yn_op_maxpool3x3s2: C must be a multiple of 4");
    launch_maxpool(x, B, H, W, C, y, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_op_shuffle_block(yn_handle* h, const char* block, const float* x, int B, int H, int W, float* y)
{
    YN_ENTER(h);
    if (!h->folded) return fail(h, "yn_op_shuffle_block before yn_fold_bn");
    h->cur = h->stream;
    const std::string P = block;
    if (!h->by_name.count(P + ".b2.pw1")) return fail(h, "unknown block '%s'", block);
    const bool s2 = h->by_name.count(P + ".b1.dw") != 0;
    const Layer& pw1 = L(h, P + ".b2.pw1");
    const int bf = pw1.cout, C = 2 * bf;
    const int Cin = s2 ? pw1.cin : C;
    const int Ho = s2 ? (H - 1) / 2 + 1 : H, Wo = s2 ? (W - 1) / 2 + 1 : W;
    const long Mi = (long)B * H * W, Mo = (long)B * Ho * Wo;
    float *t1 = nullptr, *t2 = nullptr, *tdw = nullptr, *tb1 = nullptr;
    HIPCHK(h, hipMalloc((void**)&t1, Mi * bf * sizeof(float)));
    HIPCHK(h, hipMalloc((void**)&t2, Mo * bf * sizeof(float)));
    if (s2) {
        HIPCHK(h, hipMalloc((void**)&tdw, Mo * Cin * sizeof(float)));
        HIPCHK(h, hipMalloc((void**)&tb1, Mo * bf * sizeof(float)));
        run_dw(h, L(h, P + ".b1.dw"), x, Cin, 0, B, H, W, tdw, Cin, 0);
        run_pw(h, L(h, P + ".b1.pw"), tdw, Cin, 0, Mo, tb1, bf, 0, nullptr, 0, 0);
        run_pw(h, pw1, x, Cin, 0, Mi, t1, bf, 0, nullptr, 0, 0);
        run_dw(h, L(h, P + ".b2.dw"), t1, bf, 0, B, H, W, t2, bf, 0);
        run_pw(h, L(h, P + ".b2.pw2"), t2, bf, 0, Mo, y, C, 0, tb1, bf, 0);
    } else {
        run_unit(h, P, x, B, H, W, y, t1, t2);
    }
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    (void)hipFree(t1); (void)hipFree(t2);
    if (tdw) (void)hipFree(tdw);
    if (tb1) (void)hipFree(tb1);
    return 0;
}

int yn_op_nchw_to_nhwc(yn_handle* h, const float* x, int B, int C, int H, int W, float* y)
{
    YN_ENTER(h);
    launch_nchw_to_nhwc(x, B, C, H, W, y, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

int yn_op_nhwc_to_nchw(yn_handle* h, const float* x, int B, int C, int H, int W, float* y)
{
    YN_ENTER(h);
    launch_nhwc_to_nchw(x, B, C, H, W, y, h->stream);
    HIPCHK(h, hipGetLastError());
    return 0;
}

// ---- profiling ---------------------------------------------------------------------------------
int yn_profile_enable(yn_handle* h, int enable)
{
    YN_ENTER(h);
    h->profiling = enable != 0;
    h->prof.clear();
    h->event_next = 0;
    return 0;
}

int yn_profile_count(yn_handle* h) { return h ? (int)h->prof.size() : -1; }

int yn_profile_get(yn_handle* h, int i, char* name, int name_cap, char* kernel, int kernel_cap, float* ms,
                   double* alg_flops, double* alg_bytes)
{
    YN_ENTER(h);
    if (i < 0 || i >= (int)h->prof.size()) return fail(h, "yn_profile_get: index %d out of range", i);
    const ProfRec& r = h->prof[i];
    HIPCHK(h, hipEventSynchronize(r.e1));
    float t = 0.0f;
    HIPCHK(h, hipEventElapsedTime(&t, r.e0, r.e1));
    if (name && name_cap > 0) { strncpy(name, r.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
    if (kernel && kernel_cap > 0) { strncpy(kernel, r.kernel.c_str(), kernel_cap - 1); kernel[kernel_cap - 1] = 0; }
    if (ms) *ms = t;
    if (alg_flops) *alg_flops = r.flops;
    if (alg_bytes) *alg_bytes = r.bytes;
    return 0;
}

}  // extern "C"
#pragma GCC visibility pop

#include "yn_train.inc"

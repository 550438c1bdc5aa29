// kernels_pipe.hip — unit_pipe_kernel: a stride-1 ShuffleV2 unit (backbone/shufflenetv2.py:53-63, 70-72, 14-28) as a PERSISTENT,
// software-pipelined tile walk (round 5).  Same cut and same arithmetic as unit_chain2_kernel (kernels_chain.hip):
//
//     depthwise 3x3 of the tile's pixels -> pw2 (split-f16 MFMAs) -> (x1, y) pairs = concat + shuffle -> the NEXT unit's pw1 -> global
//
// What unit_chain2_kernel could not do: its 338 workgroups of a 416 x 416 / bs 32 stage-3 unit are all resident at once and run their
// phases in lock step - a 25 MB read burst with every matrix pipe idle, ~25 k cycles of latency-bound phases with the memory system idle,
// a store burst (40 k cycles per workgroup, 0.20 of the HBM peak three rounds running).  Here
//   * a workgroup WALKS 32-row tiles (XCD-contiguous ranges, two workgroups of four wavefronts per CU), and the next tile's depthwise
//     window and pass-through rows are in flight while the current tile runs its two GEMMs: LDS-DMA (global_load_lds_dwordx4: no
//     registers, no staging instructions), issued right after the barrier that frees the buffer, waited for with a counted
//     s_waitcnt before the next tile's first barrier;
//   * the depthwise conv reads its 3 x 3 windows from that fp32 LDS image (conflict-free 16-byte reads: a pixel's channel quads are
//     consecutive lanes) - the 18-load global round trip per thread and its address arithmetic are gone;
//   * both GEMMs' weights are REGISTER-RESIDENT for the whole walk (a wavefront owns 32 columns: 8 k-steps x hi/lo x 16 bytes per
//     lane = 64 registers per matrix) - no weight traffic and no weight latency per tile (unit_chain2_kernel streamed 118 KB of
//     weights through L1 for every 64-row tile, three k-steps ahead);
//   * the pass-through half arrives in LDS too and is read where the accumulator layout needs it (16 ds_read_b32 per lane).
// Every sum runs in the order of the separate kernels (same fma chain per depthwise output, same 16-deep k-steps per accumulator):
// bit-identical to unit_chain2_kernel and to the three-kernel path (tests/test_gpu_parity.py::test_unit_chain_bit_identical_...).
#include "yn_internal.h"
#include "yn_device.h"

#include <cstdlib>
#include <type_traits>

namespace ynk {

typedef _Float16 ph16;
typedef _Float16 ph16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 ph16x4 __attribute__((ext_vector_type(4)));

// BF = branch width, compile-time: with run-time strides (bf, the plane stride, out_ld) every unrolled LDS / global access of the
// epilogues needs its own address register, and the optimiser hoists ~80 of them out of the tile loop - into the registers the weights
// live in (first form: 348 registers uncapped, 108 spilled at the 256 two workgroups per CU allow, reloaded - vmcnt(0) - inside both GEMMs).
// LAST = last unit of its stage (no next pw1: the whole shuffled row goes to global).  Both pointwise convs end in ReLU, the depthwise conv
// in nothing (every ShuffleNetV2 unit; a launch condition).
// NW = wavefronts per workgroup: 4 (two workgroups per CU) or 8 (one, twice the rows per tile - where a map is so wide that two windows do not
// fit a CU's LDS: 608 x 608 stage 3) - two wavefronts per SIMD and 256 registers either way.
template <int BF, bool LAST, int NW>
__global__ __launch_bounds__(64 * NW, 2) void unit_pipe_kernel(ChainArgs a, int tiles, float inv_w, float inv_h)
{
    constexpr int WN = BF <= 64 ? 2 : (BF <= 128 ? 4 : 8), WM = NW / WN, NT = 1, BM = 32 * WM, NTHR = 64 * NW;
    // STREAM (branches wider than 128: 15 k-steps per GEMM = 120 registers per matrix): ONE register panel.  While a GEMM walks it, every
    // k-step's fragments are replaced - behind the MFMAs that read them - by the same step of the matrix the NEXT GEMM needs (the next pointwise
    // conv of this tile, or pw2 of the next tile): a whole GEMM + an epilogue of flight time instead of unit_chain2_kernel's three k-steps
    constexpr bool STREAM = BF > 128;
    constexpr int bf = BF, KQ = (BF + 7) >> 3, PS = plane_stride(BF), S = (KQ + 1) >> 1, SMAX = S;
    // channel quads where the width allows, else pairs (58, 122: rows are 8-byte aligned only).  The window is a flat byte range either way: its
    // DMA pieces start at the 16-byte boundary below its first byte (`sh` = 0 or 8 bytes of lead-in per tile); the pass-through rows get 16-byte
    // padded LDS rows, their last piece reading up to 8 bytes past the row (and, for the tensor's last row, past the tensor: t1 / x1 are arena
    // buffers - 256-byte granules with slack behind the last one, yn_api.hip arena_take)
    constexpr int VEC = BF % 4 == 0 ? 4 : 2, CG = BF / VEC, RUN = VEC == 4 ? 4 : 8;      // channel groups; tile rows per depthwise thread
    constexpr unsigned ROWB = BF * 4u;                                  // bytes per row
    constexpr int X1C = (int)((ROWB + 15u) / 16u), X1S = X1C * 4;       // 16-byte pieces / floats per pass-through row in LDS
    typedef typename VecT<VEC>::type vec;
    constexpr bool RELU = true;
    static_assert(BF % 2 == 0 && BF <= 256 && WM >= 1 && WM * WN == NW && (VEC == 4 || !STREAM), "channel pairs at least, one 32-column tile per wavefront");
    extern __shared__ __attribute__((aligned(16))) unsigned char up_smem[];
    const int W = a.W, H = a.H, HW = H * W;
    constexpr bool last = LAST;
    constexpr int out_ld = LAST ? 2 * BF : BF;
    const unsigned win_bytes = (unsigned)(BM + 2 * W + 2) * ROWB;
    const unsigned win_lds = (win_bytes + 8u + 15u) & ~15u;             // + the lead-in, in whole pieces
    unsigned char* win = up_smem;                                       // fp32 window image: flat pixels [m0 - W - 1, m0 + BM + W + 1) x bf, from byte `sh`
    float* x1s = reinterpret_cast<float*>(up_smem + win_lds);           // pass-through rows [BM][X1S]
    ph16* Ph = reinterpret_cast<ph16*>(up_smem + win_lds + (unsigned)BM * X1S * 4u);   // operand planes [BM][PS]
    ph16* Pl = Ph + BM * PS;
    int* mtab = reinterpret_cast<int*>(Pl + BM * PS);                   // [BM] nine tap-valid bits per tile row
    float* taps = reinterpret_cast<float*>(mtab + BM);                  // depthwise weights [9][bf] + bias [bf] (registers are for the GEMM weights)
    const unsigned lds_win = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)up_smem;
    const unsigned lds_x1 = lds_win + win_lds;

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
#ifdef YN_EXP_TIMING
    const long long T_start = __builtin_readcyclecounter();
#endif
    // XCD-contiguous walk: XCD x (= blockIdx % 8) owns tiles [x TX, (x+1) TX), its workgroups take them round-robin
    const int TX = (tiles + 7) >> 3, jstep = (int)(gridDim.x >> 3);
    const int tend = ((int)(blockIdx.x & 7u) + 1) * TX < tiles ? ((int)(blockIdx.x & 7u) + 1) * TX : tiles;
    int tile = (int)(blockIdx.x & 7u) * TX + (int)(blockIdx.x >> 3);
    if (tile >= tend) return;

    // Address arithmetic of the DMA pieces: recomputed per tile from an OPAQUE copy of the thread index - as loop invariants the optimiser
    // hoists one register per piece (14 of them) out of the tile loop and spills them around the GEMMs, and a scratch reload is a vector-memory
    // load: its s_waitcnt vmcnt(0) retires the DMA pieces in flight
    const int t1_lim = ((a.M * (int)ROWB + 15) & ~15) - 16;             // last piece that holds a byte of t1 (M * bf * 8 < 2^32 is a launch condition)
    auto win_lead = [&](int tl) { return ROWB % 16u == 0 ? 0 : (((tl * BM - W - 1) * (int)ROWB) & 15); };        // 0, or 8 for channel pairs and an odd first pixel
    auto issue_window = [&](int tl) {
        int tt = t;
        asm volatile("" : "+v"(tt));
        const int g0 = (tl * BM - W - 1) * (int)ROWB;                   // first byte of the window (negative / past the end at the tensor's ends:
        const int sh = ROWB % 16u == 0 ? 0 : (g0 & 15);                 //  clamped - those pixels' taps are masked)
        const int gs = g0 - sh + tt * 16;
        const int nch = (int)((win_bytes + (unsigned)sh + 15u) >> 4);
        for (int c0 = 0; c0 < nch; c0 += NTHR) {
            int src = gs + c0 * 16;
            src = src < 0 ? 0 : (src > t1_lim ? t1_lim : src);
            if (c0 + tt < nch) dma16(a.t1, (unsigned)src, lds_win + (unsigned)(c0 + wave * 64) * 16u);
        }
    };
    auto issue_x1 = [&](int tl) {
        int tt = t;
        asm volatile("" : "+v"(tt));
        const int m0 = tl * BM;
        constexpr int nch = BM * X1C;
#pragma unroll
        for (int c0 = 0; c0 < nch; c0 += NTHR) {
            const int c = c0 + tt;
            const int row = c / X1C;
            const int j = c - row * X1C;
            const int m = m0 + row < a.M ? m0 + row : a.M - 1;
            const unsigned src = ((unsigned)m * (unsigned)a.x1_ld + (unsigned)a.x1_off) * 4u + (unsigned)j * 16u;
            if (c < nch) dma16(a.x1, src, lds_x1 + (unsigned)(c0 + wave * 64) * 16u);
        }
    };
    issue_window(tile);
    issue_x1(tile);

    // Issue order of the prologue = order of need: DMA pieces of the first tile, the depthwise taps (a few loads, bound for LDS), then the 64
    // weight-fragment loads of a lane.  The vector-memory counter retires in order, so the first wait the compiler places - for the taps, before
    // their LDS stores - is vmcnt(63): DMA and taps done, the weights still in flight under the first tile's depthwise phase (first form: one
    // vmcnt(0) in front of the loop - 12-14 k cycles of a workgroup's 25-40 k, 240 KB of fragments per CU through L2 with nothing else going on).
    constexpr int TAPQ = (10 * (BF / 2) + NTHR - 1) / NTHR;
    float2 tapv[TAPQ];
#pragma unroll
    for (int q = 0; q < TAPQ; ++q) {
        const int i = t + q * NTHR;
        const int ic = i < 10 * (BF / 2) ? i : 0;
        const int k = ic / (BF / 2), c2 = ic - k * (BF / 2);
        tapv[q] = *reinterpret_cast<const float2*>((k < 9 ? a.wdw + k * bf : a.bdw) + 2 * c2);
    }
    // ---- loop invariants: both GEMMs' B fragments of this wavefront's columns ----
    ph16x8 bw2[SMAX][NT][2], bw1[STREAM ? 1 : SMAX][NT][2];
    // No masks on the fragment loads (a masked load is USED - waited for - where it is issued): the octet past the matrix (the last half k-step
    // of an odd octet count) and the columns past Npad read the nearest valid ones instead.  Finite weights against the planes' zero K tail add
    // exact zeros; accumulator columns >= bf are never stored.
    auto load_step = [&](const void* Wh_, const void* Wl_, int s, ph16x8 (&dst)[NT][2]) {   // the fragments of k-step s
        const char* Wh = reinterpret_cast<const char*>(Wh_);
        const char* Wl = reinterpret_cast<const char*>(Wl_);
        const int kq = s * 2 + h < KQ ? s * 2 + h : KQ - 1;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int n = wn * NT * 32 + nt * 32 + l31;
            const unsigned off = ((unsigned)kq * (unsigned)a.Npad + (unsigned)(n < a.Npad ? n : a.Npad - 1)) * 16u;
            dst[nt][0] = *reinterpret_cast<const ph16x8*>(Wh + off);
            dst[nt][1] = *reinterpret_cast<const ph16x8*>(Wl + off);
        }
    };
    auto load_w = [&](const void* Wh, const void* Wl, ph16x8 (&dst)[SMAX][NT][2]) {
#pragma unroll
        for (int s = 0; s < SMAX; ++s) load_step(Wh, Wl, s, dst[s]);
    };
    // The first tile's DMA pieces and the tap loads are OLDER than the fragment loads between the two asm statements (side-effecting asm: hipcc
    // schedules no load across either), so the counted wait behind them retires exactly the pieces and the taps - in EVERY wavefront, taken or
    // not taken branches notwithstanding - and leaves the fragments in flight under the first depthwise phase.  (Round 5 relied on the wait hipcc
    // places for the tap stores: in the BF = 24 / 48 / 58 forms that one sits in a branch wavefronts 2-3 skip - ADVICE r5.)  The number of loads
    // between the markers is checked against the wait's count in the ISA by tests/test_capi_cpu.py.
    asm volatile("; YN_PIPE_FRAG_BEGIN" ::: "memory");
    load_w(a.Ws2h, a.Ws2l, bw2);
    if constexpr (!last && !STREAM) load_w(a.Ws1h, a.Ws1l, bw1);
    asm volatile("; YN_PIPE_FRAG_END\n\ts_waitcnt vmcnt(%0)" ::"i"(SMAX * NT * 2 * ((!last && !STREAM) ? 2 : 1)) : "memory");
    float bias2[NT], bias1n[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int n = wn * NT * 32 + nt * 32 + l31;
        bias2[nt] = n < bf ? a.b2[n] : 0.0f;
        bias1n[nt] = (!last && n < bf) ? a.b1n[n] : 0.0f;
    }
    static_assert(CG * (BM / RUN) <= NTHR, "one depthwise round per tile");     // channel groups x runs of RUN tile rows: the depthwise phase's threads
    const int cq = t % CG, pl = t / CG;
    const bool worker = pl < BM / RUN;
    {
#pragma unroll
        for (int q = 0; q < TAPQ; ++q) {
            const int i = t + q * NTHR;
            if (i < 10 * (BF / 2)) *reinterpret_cast<float2*>(taps + 2 * i) = tapv[q];
        }
    }
    // K tail of both planes: zero once (nothing below writes columns >= bf)
    {
        const int padn = PS - bf;
        for (int i = t; i < BM * padn; i += NTHR) { const int r = i / padn, c2 = bf + i - r * padn; Ph[r * PS + c2] = (ph16)0.0f; Pl[r * PS + c2] = (ph16)0.0f; }
    }
    float amax = 0.0f;                                                  // range guard (yn_device.h)
    const int jhi = bf >> 1;
    f32x16 acc0[NT], acc1[NT];

    // activation of the pointwise convs: one v_max_f32 in the RELU instantiation (apply_act's NaN -> 0 and -0 -> +0 included), the run-time form otherwise
    auto act1 = [&](float v, int act) { return RELU ? __builtin_fmaxf(v, 0.0f) : apply_act(v, act); };
    // refill != null (STREAM): k-step s of that matrix replaces the fragments the MFMAs of step s have just read
    const unsigned lane_w = ((unsigned)h * (WN * 32u) + (unsigned)wn * 32u + (unsigned)l31) * 16u;      // byte offset of this lane's fragment inside a k-step of a pack
    // which: 0 = no refill, 1 = refill with the next pointwise conv's matrix, 2 = with pw2 (for the next tile); do_refill: wave-uniform
    auto gemm = [&](ph16x8 (&bw)[SMAX][NT][2], auto which, bool refill) {
        const char* refill_h = reinterpret_cast<const char*>(decltype(which)::value == 1 ? a.Ws1h : a.Ws2h);
        const char* refill_l = reinterpret_cast<const char*>(decltype(which)::value == 1 ? a.Ws1l : a.Ws2l);
        unsigned lw = lane_w;                                           // (opaque per call: as loop invariants the per-step addresses are hoisted out of the tile loop)
        asm volatile("" : "+v"(lw));
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) { acc0[i][k] = 0.0f; acc1[i][k] = 0.0f; }
        const ph16* ahp = Ph + (wm * 32 + l31) * PS + h * 8;
        const ph16* alp = Pl + (wm * 32 + l31) * PS + h * 8;
#pragma unroll
        for (int s = 0; s < SMAX; ++s) {
            const ph16x8 ah = *reinterpret_cast<const ph16x8*>(ahp + s * 16);
            const ph16x8 al = *reinterpret_cast<const ph16x8*>(alp + s * 16);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                acc0[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bw[s][nt][0], acc0[nt], 0, 0, 0);
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bw[s][nt][1], acc1[nt], 0, 0, 0);
                acc1[nt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bw[s][nt][0], acc1[nt], 0, 0, 0);
            }
            if constexpr (STREAM && decltype(which)::value != 0) {
                // No mask on this path (a masked load needs a temporary per load in flight: 120 registers), and ONE lane offset for all k-steps
                // (wave-uniform base + s * 8 KB: per-step 64-bit lane addresses are 60 loop-invariant registers - hoisted, spilled and reloaded
                // through vmcnt(0) in the first form).  The one octet past the matrix (the last half k-step of an odd octet count) reads the last
                // valid octet instead: finite weights against the planes' zero K tail.
                if (refill) {
                    const unsigned kq = (s * 2 + 1 < KQ) ? (unsigned)(s * 2 + h) : (unsigned)(s * 2 + h < KQ ? s * 2 + h : KQ - 1);
                    const unsigned off = (s * 2 + 1 < KQ) ? lw + (unsigned)s * (2u * WN * 32u * 16u) : lw - (unsigned)h * (WN * 32u * 16u) + kq * (WN * 32u * 16u);
                    bw[s][0][0] = *reinterpret_cast<const ph16x8*>(refill_h + off);
                    bw[s][0][1] = *reinterpret_cast<const ph16x8*>(refill_l + off);
                }
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc0[nt][r] = __builtin_fmaf(acc1[nt][r], 1.0f / 2048.0f, acc0[nt][r]);
    };
    // STREAM: hipcc is TOLD where the streamed panel has landed (an empty asm that reads and writes its 30 fragment registers: its own wait for the
    // refill loads goes HERE - behind the hand-placed drain - instead of in front of the next tile's first GEMM's
    // MFMAs with the count of the loads it knows of, vmcnt(0) at the last k-step, which retires the DMA pieces issued just before that GEMM)
    auto panel_landed = [&]() {
        if constexpr (STREAM) {
            static_assert(!STREAM || (SMAX == 15 && NT == 1), "30 asm operands");
#define YN_P2(s) "+v"(bw2[s][0][0]), "+v"(bw2[s][0][1])
            asm volatile("" : YN_P2(0), YN_P2(1), YN_P2(2), YN_P2(3), YN_P2(4), YN_P2(5), YN_P2(6), YN_P2(7), YN_P2(8), YN_P2(9), YN_P2(10), YN_P2(11), YN_P2(12), YN_P2(13), YN_P2(SMAX - 1));
#undef YN_P2
        }
    };
    auto split2 = [&](int r, int c, float v0, float v1) {               // two adjacent channels of row r -> both planes
        amax = range_track(range_track(amax, v0), v1);
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 hi, lo;
        hi[0] = (ph16)v0; hi[1] = (ph16)v1;
        lo[0] = (ph16)((v0 - (float)hi[0]) * 2048.0f); lo[1] = (ph16)((v1 - (float)hi[1]) * 2048.0f);
        *reinterpret_cast<h2*>(Ph + r * PS + c) = hi;
        *reinterpret_cast<h2*>(Pl + r * PS + c) = lo;
    };

    // tap-valid bits of a tile's rows (zero padding of the 3 x 3 window; idle rows: nothing valid) -> mtab; written one tile ahead, behind the
    // barrier that ends the depthwise phase (the only reader)
    auto write_mtab = [&](int tl) {
        int tq = threadIdx.x;                                           // (opaque: as a loop invariant the mtab address is one more register held across the GEMMs -
        asm volatile("" : "+v"(tq));                                    //  the eight-wavefront form of width 116 spilled it, and a scratch reload's vmcnt(0) retires the DMA pieces in flight)
        if (tq < BM) {
            const int t = tq;
            const int m0 = tl * BM;
            const int rem0 = m0 % HW;                                   // wave-uniform
            const int y0 = rem0 / W, x0 = rem0 - y0 * W;
            const int q = x0 + t;
            const int dy = (int)(((float)q + 0.5f) * inv_w);            // q / W (exact: q < 2^16)
            const int x = q - dy * W;
            const int yy = y0 + dy;
            const int y = yy - (int)(((float)yy + 0.5f) * inv_h) * H;   // rows past the image's last one continue in the next image
            const int yb = (y >= 1 ? 1 : 0) | 2 | (y + 1 < H ? 4 : 0), xb = (x >= 1 ? 1 : 0) | 2 | (x + 1 < W ? 4 : 0);
            int bits = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
                if ((yb >> ky) & 1) bits |= xb << (3 * ky);
            mtab[t] = m0 + t < a.M ? bits : 0;
        }
    };
    write_mtab(tile);
#ifdef YN_EXP_TIMING
    const long long T_loop = __builtin_readcyclecounter();
    long long TS[12]; int tsn = 0, titer = 0;
#define YN_TS() TS[tsn++] = __builtin_readcyclecounter()
#else
#define YN_TS()
#endif
    // One tile.  Instantiated TWICE: the workgroup's first tile ahead of the loop, the rest in it.  hipcc flushes the vector-memory counter in
    // front of a loop whose body uses loaded values but holds no loads it counts (the DMA pieces are invisible to it) - with the first tile
    // inside the loop that flush is a vmcnt(0) on 64 weight-fragment loads per lane before the first depthwise phase may start (12-14 k cycles of
    // a workgroup's 25-40 k).  Peeled, the waits sit where the first tile USES the fragments: in front of its two GEMMs.
    auto do_tile = [&](const int tile, const int next) __attribute__((always_inline)) {
#ifdef YN_EXP_TIMING
        tsn = 0;
#endif
        YN_TS();
        const int m0 = tile * BM;
        const int nrows = a.M - m0 < BM ? a.M - m0 : BM;
        YN_TS();
        lds_barrier();      // (1) this tile's window and pass-through rows have landed (every wavefront waited for its pieces before it got here), mtab is written

        YN_TS();
        // ---- depthwise 3x3 from the LDS window -> split planes (the fma chain of dwconv3x3_kernel).  Thread = one channel group (quad, or pair)
        //      of a RUN of consecutive tile rows (4, or 8): its 3 x (RUN + 2) window is read row by row (18 sixteen-byte LDS reads for four
        //      outputs instead of 36) and the outputs' fma chains are independent - one round per tile instead of four dependent ones ----
        if (worker) {
            vec wd[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wd[k] = *reinterpret_cast<const vec*>(taps + k * bf + VEC * cq);
            const vec bd = *reinterpret_cast<const vec*>(taps + 9 * bf + VEC * cq);
            const int r0 = RUN * pl;
            int bits[RUN];
#pragma unroll
            for (int q = 0; q < RUN / 4; ++q) {
                const int4 b4 = *reinterpret_cast<const int4*>(mtab + r0 + 4 * q);
                bits[4 * q] = b4.x; bits[4 * q + 1] = b4.y; bits[4 * q + 2] = b4.z; bits[4 * q + 3] = b4.w;
            }
            // window pixel r0 = the top-left neighbour of tile row r0; the image starts `lead` bytes into the buffer
            const unsigned char* wp = win + win_lead(tile) + ((unsigned)r0 * bf + VEC * cq) * 4u;
            vec acc[RUN];
#pragma unroll
            for (int i = 0; i < RUN; ++i) acc[i] = bd;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                vec row[RUN + 2];
#pragma unroll
                for (int i = 0; i < RUN + 2; ++i) row[i] = *reinterpret_cast<const vec*>(wp + (unsigned)((ky * W + i) * bf) * 4u);
#pragma unroll
                for (int i = 0; i < RUN; ++i)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const bool ok = (bits[i] >> (ky * 3 + kx)) & 1;
                        vec v = row[i + kx];
                        if constexpr (VEC == 4) v = make_float4(ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f);
                        else v = make_float2(ok ? v.x : 0.0f, ok ? v.y : 0.0f);
                        vfma(acc[i], v, wd[ky * 3 + kx]);
                    }
            }
#pragma unroll
            for (int i = 0; i < RUN; ++i) {
                float o[VEC];
                if constexpr (VEC == 4) { o[0] = acc[i].x; o[1] = acc[i].y; o[2] = acc[i].z; o[3] = acc[i].w; }
                else { o[0] = acc[i].x; o[1] = acc[i].y; }
                typedef _Float16 hv __attribute__((ext_vector_type(VEC)));
                hv hi, lo;
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    // the fp32 value is the result: without this the optimiser folds the last fma and the conversion below into v_fma_mixlo_f16 -
                    // ONE rounding, straight to f16 - and a value that lies exactly between two f16 neighbours once the fma has been rounded to fp32
                    // gets the other hi (hi + lo * 2^-11 is the same number either way; bit-identity with the other kernels is not: 1 pixel of 200)
                    asm volatile("" : "+v"(o[j]));
                    amax = range_track(amax, o[j]);
                    hi[j] = (ph16)o[j];
                    lo[j] = (ph16)((o[j] - (float)hi[j]) * 2048.0f);
                }
                *reinterpret_cast<hv*>(Ph + (r0 + i) * PS + VEC * cq) = hi;
                *reinterpret_cast<hv*>(Pl + (r0 + i) * PS + VEC * cq) = lo;
            }
        }
        YN_TS();
        lds_barrier();      // (2) planes complete; every window read has returned: the window buffer is free
        YN_TS();
        if (next >= 0) { issue_window(next); write_mtab(next); }        // in flight under both GEMMs and the first epilogue
        YN_TS();
        if constexpr (STREAM) {                                         // ... and, behind its k-steps, the matrix of the next GEMM
            if constexpr (last) gemm(bw2, std::integral_constant<int, 0>{}, false);     // (its own panel: stays)
            else gemm(bw2, std::integral_constant<int, 1>{}, true);
        } else {
            gemm(bw2, std::integral_constant<int, 0>{}, false);
        }
        YN_TS();
        if constexpr (!last) lds_barrier();   // (3) every wavefront is done reading the planes (the epilogue writes x2' into them)

        // ---- y = act(acc + b2) straight to its final place: (x1, y) pairs -> global, or split into the planes as x2'.  Row groups of
        //      eight are live or idle as a whole (M % 8 == 0 is a launch condition): scalar branches only, one exec region per destination ----
        {
            const __amdgpu_buffer_rsrc_t out_rs = buf_rsrc(a.out + (size_t)m0 * out_ld);       // (one lane offset + immediates instead of a 64-bit lane address per row)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const int n = wn * NT * 32 + nt * 32 + l31;
                const float bias = bias2[nt];
                const bool to_global = n < (last ? bf : jhi);
                const bool to_plane = !last && n >= jhi && n < bf;
                const float* xr = x1s + (wm * 32 + 4 * h) * X1S + (n < bf ? n : 0);
#pragma unroll
                for (int g8 = 0; g8 < 4; ++g8) {
                    if (wm * 32 + 8 * g8 < nrows) {                     // scalar condition
                        float y[4], xv[4];
#pragma unroll
                        for (int q = 0; q < 4; ++q) xv[q] = xr[(q + 8 * g8) * X1S];
#pragma unroll
                        for (int q = 0; q < 4; ++q) y[q] = act1(acc0[nt][4 * g8 + q] + bias, a.act2);
                        if (to_global) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const int row = wm * 32 + q + 8 * g8 + 4 * h;
                                buf_store_b64<0>(out_rs, (unsigned)(row * out_ld + 2 * n) * 4u, make_float2(xv[q], y[q]));
                            }
                        }
                        if (to_plane) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) split2(wm * 32 + q + 8 * g8 + 4 * h, 2 * (n - jhi), xv[q], y[q]);
                        }
                    }
                }
            }
        }
        YN_TS();
        lds_barrier();      // (4) x2' complete; the pass-through buffer is free
        if (next >= 0) issue_x1(next);
        YN_TS();
        if constexpr (!last) {
            // ---- the next unit's pw1 on x2' -> global ----
            if constexpr (STREAM) gemm(bw2, std::integral_constant<int, 2>{}, next >= 0);     // bw2 holds pw1n now; pw2 of the next tile follows
            else gemm(bw1, std::integral_constant<int, 0>{}, false);
            YN_TS();
            vm_drain();
            panel_landed();
            YN_TS();     // the next tile's pieces (and this tile's first stores, long gone) - BEFORE the stores below, which need not be waited for
            GemmArgs e{};
            e.out = a.t1n; e.out_ld = bf; e.out_off = 0; e.M = m0 + nrows; e.N = bf; e.Npad = a.Npad; e.bias = a.b1n; e.act = a.act1n; e.pass = nullptr;
            gemm_epilogue<NT>(e, acc0, m0 + wm * 32, wn * NT * 32, VEC == 4, lane, bias1n);
        } else {
            vm_drain();
        }
#ifdef YN_EXP_TIMING
        YN_TS();
        if (!last && t == 0 && (blockIdx.x % 61) == 5)
            printf("pipe bf %d blk %d iter %d: top %lld wait1 %lld dw %lld bar2 %lld issue %lld gemm1 %lld bar3+epi1 %lld bar4+x1 %lld gemm2 %lld drain %lld epi2 %lld total %lld\n", bf, (int)blockIdx.x, titer,
                   TS[1] - TS[0], TS[2] - TS[1], TS[3] - TS[2], TS[4] - TS[3], TS[5] - TS[4], TS[6] - TS[5], TS[7] - TS[6], TS[8] - TS[7], TS[9] - TS[8], TS[10] - TS[9], TS[11] - TS[10], TS[11] - TS[0]);
        ++titer;
#endif
    };
    {
        int next = tile + jstep < tend ? tile + jstep : -1;
        do_tile(tile, next);
        while (next >= 0) {
            tile = next;
            next = tile + jstep < tend ? tile + jstep : -1;
            do_tile(tile, next);
        }
    }
#undef YN_TS
    range_report(a.ovf, amax);
#ifdef YN_EXP_TIMING
    if (t == 0 && (blockIdx.x % 37) == 3)
        printf("pipewg bf %d last %d blk %d tiles %d prologue %lld total %lld start-stamp %lld\n", bf, (int)last, (int)blockIdx.x, titer, T_loop - T_start, __builtin_readcyclecounter() - T_start, T_start & 0xffffff);
#endif
}

static size_t unit_pipe_lds(int bf, int W, int BM)
{
    const size_t win = (((size_t)(BM + 2 * W + 2) * bf * 4 + 8 + 15) & ~(size_t)15), x1row = (((size_t)bf * 4 + 15) / 16) * 16;
    return win + (size_t)BM * x1row + (size_t)2 * BM * plane_stride(bf) * 2 + (size_t)BM * 4 + (size_t)10 * bf * 4;
}

// The persistent form of a stride-1 unit where it applies: split-f16 family, an instantiated branch width, ReLU pointwise / linear depthwise
// convs, dense depthwise input, whole row groups of eight, a workgroup small enough for two per CU, 32-bit byte offsets, enough tiles to
// walk.  false = not launched (the caller runs unit_chain2_kernel).
bool launch_unit_pipe(const ChainArgs& a, hipStream_t s, bool dry)
{
    // the form is the handle's choice (yn_chain_pipe -> ChainArgs::pipe_mode: 0 by the size rule, 1 never, 2 also for few tiles); the A/B knobs
    // below are read once per process
    const int mode = a.pipe_mode == 1 ? 0 : (a.pipe_mode == 2 ? 2 : 1);
    // size rule: the walk pays from about one tile per workgroup slot; the streamed wide form (a workgroup's whole-panel prefetch instead of three
    // k-steps of look-ahead) at every size: one 608 x 608 image 0.652 -> 0.639 ms
    static const int min_tiles_env = getenv("YN_CHAIN_PIPE_MIN") ? atoi(getenv("YN_CHAIN_PIPE_MIN")) : -1;
    const int min_tiles = min_tiles_env >= 0 ? min_tiles_env : (a.bf > 128 ? 1 : 256);
    // walking workgroups: a multiple of 8 (XCD-contiguous ranges: jstep = grid / 8 tiles per step - a cap below 8 would never advance), at least 16 so
    // that the eight-wavefront forms' half is one too
    static const int wg_cap = [] { int c = getenv("YN_CHAIN_PIPE_G") ? atoi(getenv("YN_CHAIN_PIPE_G")) : 512; c &= ~15; return c < 16 ? 16 : c; }();
    const bool last = a.Wp1n == nullptr;
    static const bool force8 = getenv("YN_CHAIN_PIPE_NW") && atoi(getenv("YN_CHAIN_PIPE_NW")) == 8;      // A/B: the eight-wavefront form also where two windows fit
    if (!mode || !a.Ws2h || (!last && !a.Ws1h)) return false;
    if (a.dw_act != 0 || a.act2 != 1 || (!last && a.act1n != 1)) return false;
    if ((a.M & 7) || a.t1_ld != a.bf || a.t1_off != 0 || ((a.x1_ld | a.x1_off) & ((a.bf & 3) ? 1 : 3)) || a.out_ld != (last ? 2 * a.bf : a.bf)) return false;
    if (a.Npad != ((a.bf + 31) & ~31)) return false;
    if (a.bf > 128 && a.Npad != 256) return false;                       // the streamed form indexes its pack with the full eight column tiles
    if ((double)a.M * a.bf * 8.0 >= 4.0e9 || (double)a.M * a.x1_ld * 4.0 >= 4.0e9) return false;
#define YN_UP(BFv, LASTv, NWv)                                                                                           \
    {                                                                                                                    \
        constexpr int BM = 32 * (NWv / (BFv <= 64 ? 2 : (BFv <= 128 ? 4 : 8)));                                                             \
        const size_t lds = unit_pipe_lds(a.bf, a.W, BM);                                                                 \
        const int tiles = (a.M + BM - 1) / BM;                                                                           \
        if (lds <= (size_t)(NWv == 4 ? 80 : 160) * 1024 && !(NWv == 4 && force8)) {                                      \
            if (mode < 2 && tiles < min_tiles) return false;                                                             \
            if (dry) return true;                                                                                        \
            static unsigned long long attr = 0;                                                                          \
            if (attr_pending(attr)) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(unit_pipe_kernel<BFv, LASTv, NWv>), \
                                                   hipFuncAttributeMaxDynamicSharedMemorySize, (NWv == 4 ? 80 : 160) * 1024); }  \
            unsigned g = xcd_grid((unsigned)tiles);                                                                      \
            const unsigned cap = (unsigned)(NWv == 4 ? wg_cap : wg_cap / 2);                                             \
            if (g > cap) g = cap;                                                                                        \
            set_last_kernel_name("unit_pipe_kernel<" #BFv "," #LASTv "," #NWv ">");                                       \
            hipLaunchKernelGGL((unit_pipe_kernel<BFv, LASTv, NWv>), dim3(g), dim3(64 * NWv), lds, s, a, tiles, 1.0f / (float)a.W, 1.0f / (float)a.H); \
            return true;                                                                                                 \
        }                                                                                                                \
    }
    // two four-wavefront workgroups per CU where their windows fit, else one of eight
#define YN_UPB(BFv) if (a.bf == BFv) { if (last) { YN_UP(BFv, true, 4) YN_UP(BFv, true, 8) } else { YN_UP(BFv, false, 4) YN_UP(BFv, false, 8) } return false; }
#define YN_UPW(BFv) if (a.bf == BFv) { if (last) { YN_UP(BFv, true, 8) } else { YN_UP(BFv, false, 8) } return false; }        // eight column tiles: eight wavefronts
    YN_UPW(232)          // 1.0x stage 4 (streamed weights)
    YN_UPB(116)          // 1.0x stage 3
    YN_UPB(58)           // 1.0x stage 2 (channel pairs)
    YN_UPB(96)           // 0.5x stage 4
    YN_UPB(48)           // 0.5x stage 3
    YN_UPB(24)           // 0.5x stage 2
#undef YN_UPB
#undef YN_UPW
#undef YN_UP
    return false;
}


// -------------------------------------------------------------------------------------------------
// pw_pipe_kernel: a pointwise conv (utils/modules.py:8-18 folded; backbone/shufflenetv2.py:42-44, 56-58: the first 1x1 conv of a unit's
// branch 2) as the same persistent walk - unit_pipe_kernel without its depthwise conv and second GEMM.  gemm_split_kernel re-streams the
// layer's whole weight matrix for every 128-row tile (K = N = 116: 59 KB of fragments against 59 KB of activations read) through a
// register -> LDS staging pass with a barrier pair per K chunk; here a workgroup walks 32-row (K <= 64: 64-row) tiles with the matrix
// REGISTER-resident (64 registers), the next tile's fp32 rows arrive by LDS-DMA under the current tile's split pass, GEMM and stores, and
// 120 registers / 33 KB of LDS leave room for FOUR workgroups per CU.  Same 16-deep k-steps in sequence, same epilogue function:
// bit-identical to every gemm_split_kernel configuration; the autotuner times it next to them (the last pointwise configuration index).
// -------------------------------------------------------------------------------------------------
template <int KK, int NPAD, int NW, int OCC>
__global__ __launch_bounds__(64 * NW, OCC) void pw_pipe_kernel(GemmArgs a, int tiles)
{
    // NW = 4: up to 128 output columns, 128 registers, four workgroups per CU.  NW = 8 (K = 232, the stage-4 width): 256 columns - one 32-column
    // strip of B per wavefront is 120 registers of fragments - on one workgroup per CU.
    constexpr int WN = NPAD <= 32 ? 1 : (NPAD <= 64 ? 2 : (NPAD <= 128 ? 4 : 8)), WM = NW / WN, BM = 32 * WM, NTHR = 64 * NW;
    constexpr int KQ = (KK + 7) >> 3, PS = plane_stride(KK), S = (KQ + 1) >> 1;
    constexpr int VEC = KK % 4 == 0 ? 4 : 2, CG = KK / VEC;
    constexpr unsigned ROWB = KK * 4u;
    constexpr int X1C = (int)((ROWB + 15u) / 16u), X1S = X1C * 4;       // 16-byte pieces / floats per input row in LDS
    typedef typename VecT<VEC>::type vec;
    static_assert(KK % 2 == 0 && S <= (NW * OCC == 16 ? 8 : 15) && NPAD <= 32 * NW && WM >= 1, "register-resident B fragments");
    extern __shared__ __attribute__((aligned(16))) unsigned char pp_smem[];
    float* raw = reinterpret_cast<float*>(pp_smem);                     // [BM][X1S] fp32 input rows
    ph16* Ph = reinterpret_cast<ph16*>(pp_smem + (unsigned)BM * X1S * 4u);      // operand planes [BM][PS]
    ph16* Pl = Ph + BM * PS;
    const unsigned lds_raw = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)pp_smem;

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    const int TX = (tiles + 7) >> 3, jstep = (int)(gridDim.x >> 3);
    const int tend = ((int)(blockIdx.x & 7u) + 1) * TX < tiles ? ((int)(blockIdx.x & 7u) + 1) * TX : tiles;
    int tile = (int)(blockIdx.x & 7u) * TX + (int)(blockIdx.x >> 3);
    if (tile >= tend) return;

    auto issue_rows = [&](int tl) {
        int tt = t;
        asm volatile("" : "+v"(tt));
        const int m0 = tl * BM;
        constexpr int nch = BM * X1C;
#pragma unroll
        for (int c0 = 0; c0 < nch; c0 += NTHR) {
            const int c = c0 + tt;
            const int row = c / X1C;
            const int j = c - row * X1C;
            const int m = m0 + row < a.M ? m0 + row : a.M - 1;
            const unsigned src = ((unsigned)m * (unsigned)a.in_ld + (unsigned)a.in_off) * 4u + (unsigned)j * 16u;
            if (c < nch) dma16(a.in, src, lds_raw + (unsigned)(c0 + wave * 64) * 16u);
        }
    };
    issue_rows(tile);
    ph16x8 bw[S][2];
    {
        const int n = wn * 32 + l31;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const int kq = s * 2 + h < KQ ? s * 2 + h : KQ - 1;        // (no masks: unit_pipe_kernel's load_step)
            const unsigned off = ((unsigned)kq * (unsigned)a.Npad + (unsigned)(n < a.Npad ? n : a.Npad - 1)) * 16u;
            bw[s][0] = *reinterpret_cast<const ph16x8*>(reinterpret_cast<const char*>(a.Wsh) + off);
            bw[s][1] = *reinterpret_cast<const ph16x8*>(reinterpret_cast<const char*>(a.Wsl) + off);
        }
    }
    float bias1[1];
    { const int n = wn * 32 + l31; bias1[0] = n < a.N ? a.bias[n] : 0.0f; }
    {
        constexpr int padn = PS - KK;
        for (int i = t; i < BM * padn; i += NTHR) { const int r = i / padn, c2 = KK + i - r * padn; Ph[r * PS + c2] = (ph16)0.0f; Pl[r * PS + c2] = (ph16)0.0f; }
    }
    float amax = 0.0f;
    const bool vecO = ((a.N | a.out_ld | a.out_off) & 3) == 0;

    auto do_tile = [&](const int tile, const int next) __attribute__((always_inline)) {
        const int m0 = tile * BM;
        lds_barrier();      // (1) this tile's rows have landed (every wavefront waited for its pieces), the previous tile's GEMM is done with the planes
        // ---- split pass: fp32 rows -> hi / lo planes ----
        {
            int tt = threadIdx.x;
            asm volatile("" : "+v"(tt));
#pragma unroll
            for (int i0 = 0; i0 < BM * CG; i0 += NTHR) {
                const int i = i0 + tt;
                if (i < BM * CG) {
                    const int r = i / CG, cq = i - r * CG;
                    const vec v = *reinterpret_cast<const vec*>(raw + r * X1S + VEC * cq);
                    float o[VEC];
                    if constexpr (VEC == 4) { o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w; } else { o[0] = v.x; o[1] = v.y; }
                    typedef _Float16 hv __attribute__((ext_vector_type(VEC)));
                    hv hi, lo;
#pragma unroll
                    for (int j = 0; j < VEC; ++j) { amax = range_track(amax, o[j]); hi[j] = (ph16)o[j]; lo[j] = (ph16)((o[j] - (float)hi[j]) * 2048.0f); }
                    *reinterpret_cast<hv*>(Ph + r * PS + VEC * cq) = hi;
                    *reinterpret_cast<hv*>(Pl + r * PS + VEC * cq) = lo;
                }
            }
        }
        lds_barrier();      // (2) planes complete; the row buffer is free
        if (next >= 0) issue_rows(next);
        f32x16 acc0[1], acc1[1];
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc0[0][k] = 0.0f; acc1[0][k] = 0.0f; }
        {
            int ll = threadIdx.x & 63;
            asm volatile("" : "+v"(ll));
            const ph16* ahp = Ph + (wm * 32 + (ll & 31)) * PS + (ll >> 5) * 8;
            const ph16* alp = Pl + (wm * 32 + (ll & 31)) * PS + (ll >> 5) * 8;
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const ph16x8 ah = *reinterpret_cast<const ph16x8*>(ahp + s * 16);
                const ph16x8 al = *reinterpret_cast<const ph16x8*>(alp + s * 16);
                acc0[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bw[s][0], acc0[0], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bw[s][1], acc1[0], 0, 0, 0);
                acc1[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bw[s][0], acc1[0], 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[0][r] = __builtin_fmaf(acc1[0][r], 1.0f / 2048.0f, acc0[0][r]);
        vm_drain();         // the next tile's pieces - BEFORE this tile's stores, which need not be waited for
        gemm_epilogue_impl<1, false>(a, acc0, m0 + wm * 32, wn * 32, vecO, threadIdx.x & 63, bias1);     // (no pass-through form: its loads would make hipcc drain the memory counter - stores included - at the top of every tile)
    };
    {
        int next = tile + jstep < tend ? tile + jstep : -1;
        vm_drain();         // the first tile's pieces (and the fragments): nothing else in the prologue waits for them
        do_tile(tile, next);
        while (next >= 0) {
            tile = next;
            next = tile + jstep < tend ? tile + jstep : -1;
            do_tile(tile, next);
        }
    }
    range_report(a.ovf, amax);
}

// false = this layer has no instantiated form (the caller takes a gemm_split_kernel configuration)
bool launch_pw_pipe(const GemmArgs& a, hipStream_t s)
{
    if (!a.Wsh || !a.Wsl || a.pass || (a.in_ld & 1) || (a.in_off & 1) || a.Npad != ((a.N + 31) & ~31) || a.M < 64) return false;
    if ((double)a.M * a.in_ld * 4.0 >= 4.0e9) return false;
    if ((a.K & 3) && a.in_slack < 8) return false;          // the last 16-byte piece of a row overruns it by 8 bytes: only where the caller vouches for the tensor's end
    static const int wg_cap = [] { int c = getenv("YN_PW_PIPE_G") ? atoi(getenv("YN_PW_PIPE_G")) : 1024; c &= ~7; return c < 8 ? 8 : c; }();    // a multiple of 8, at least 8 (jstep = grid / 8)
#define YN_PP(Kv, Nv, NWv, OCCv)                                                                                             \
    if (a.K == Kv && a.Npad == Nv) {                                                                                     \
        constexpr int BM = 32 * (NWv / (Nv <= 32 ? 1 : (Nv <= 64 ? 2 : (Nv <= 128 ? 4 : 8))));                            \
        const int tiles = (a.M + BM - 1) / BM;                                                                           \
        const size_t lds = (size_t)BM * (((size_t)Kv * 4 + 15) / 16) * 16 + (size_t)2 * BM * plane_stride(Kv) * 2;        \
        unsigned g = xcd_grid((unsigned)tiles);                                                                          \
        if (g > (unsigned)wg_cap) g = (unsigned)wg_cap;                                                                  \
        if (g > 256u * OCCv) g = 256u * OCCv;                                                                            \
        set_last_kernel_name("pw_pipe_kernel<" #Kv "," #Nv ">");                                                         \
        hipLaunchKernelGGL((pw_pipe_kernel<Kv, Nv, NWv, OCCv>), dim3(g), dim3(64 * NWv), lds, s, a, tiles);                    \
        return true;                                                                                                     \
    }
    YN_PP(116, 128, 4, 4) YN_PP(116, 96, 4, 4) YN_PP(58, 64, 4, 4) YN_PP(96, 96, 4, 4) YN_PP(48, 64, 4, 4) YN_PP(24, 32, 4, 4) YN_PP(24, 64, 4, 4)
    YN_PP(232, 256, 8, 1) YN_PP(232, 96, 4, 2)
#undef YN_PP
    return false;
}

}  // namespace ynk

// kernels_stage.hip — stage_pipe_kernel: the stride-1 ShuffleV2 units of one backbone stage (backbone/shufflenetv2.py:53-78, 118-125) as ONE
// persistent launch (round 6).  unit_pipe_kernel (kernels_pipe.hip) runs one unit per launch: every launch pays a weight prologue (131 KB of
// fragments per workgroup through L2: a third of a workgroup's life at 1.3 tiles per workgroup), a 2-tile critical path for 1.32 tiles of
// average work and a kernel boundary.  Here the units of a stage share one launch:
//
//   * WORK ITEMS are (unit, 32-row tile) pairs, handed out by TICKET: eight queues (one per eighth of the tile range = per XCD, chosen by the
//     workgroup's HW_REG_XCC_ID - a speed hint only: an exhausted queue's workgroups take tickets of the next one), each in unit-major order.
//     Item (u, T) needs the outputs of (u-1, T-1), (u-1, T), (u-1, T+1) (the 3 x 3 window's halo) - all EARLIER tickets of their queues, so
//     whoever holds them is resident and running: no wait on a workgroup that has not been dispatched, whatever else shares the chip
//     (the four-stream run) and whatever the placement.
//   * HAND-OFF between workgroups follows the placement-independent protocol (cdna_hip_programming.md 6, Guideline 16, form R1): the
//     producer stores its rows WRITE-THROUGH (sc1), every storing wavefront drains its vector-memory counter, a barrier, ONE lane stores the
//     tile's ready flag (sc1); the consumer polls the three flags relaxed (sc1 loads, one wavefront) and reads the rows with sc1 LDS-DMA
//     pieces (L1 bypassed, L2-served).  Every spin is bounded (a wall-clock limit raises the handle's timeout word).
//   * a tile's flags are polled while the PREVIOUS tile's depthwise phase runs: ready -> the window's DMA pieces are issued under the two GEMMs
//     as in unit_pipe_kernel; not yet -> polled again under the first GEMM (pieces issued at the fourth barrier); still not -> a bounded wait
//     at the end of the tile.
//   * the two weight matrices of the NEXT item's unit replace the register panels behind the MFMAs that have just read them (the STREAM
//     mechanism of unit_pipe_kernel<232>), its depthwise taps and biases arrive by LDS-DMA (taps: single buffer, free after the depthwise
//     phase; biases: double-buffered) - no prologue per unit.
//   * the polled words (queue heads, flags, exit count) are zero between launches: the last workgroup to leave resets them (they are zeroed
//     once when the handle allocates them) - no memset node per launch, legal under graph replay.
// Arithmetic: the tile body is unit_pipe_kernel's (same fma chains, same k order, same epilogues): bit-identical to the per-unit launches.
// The LAST unit of a stage (whole shuffled rows to global, no next pw1) stays a unit_pipe_kernel launch.
#include "yn_internal.h"
#include "yn_device.h"

#include <cstdlib>

namespace ynk {

typedef _Float16 sh16;
typedef _Float16 sh16x8 __attribute__((ext_vector_type(8)));
typedef float sf32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) unsigned gu32s;

#define YN_RLX_AGENT __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT

// LDS-DMA piece with sc1: bypasses this CU's L1 (another workgroup's write-through rows are read from L2 / memory, never from a stale L1 line)
__device__ __forceinline__ void dma16_sc1(const void* gbase, unsigned goff, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(goff), "s"(gbase), "s"(lds_dst)
                 : "memory");
}
template <int BF, int NW, bool PUB_EARLY>
__global__ __launch_bounds__(64 * NW, 2) void stage_pipe_kernel(StageArgs a)
{
    constexpr int WN = BF <= 64 ? 2 : 4, WM = NW / WN, BM = 32 * WM, NTHR = 64 * NW;
    constexpr int KQ = (BF + 7) >> 3, PS = plane_stride(BF), S = (KQ + 1) >> 1, NPAD = (BF + 31) & ~31;
    constexpr int CG = BF / 4, RUN = 4;
    constexpr unsigned ROWB = BF * 4u;
    constexpr int X1C = BF / 4, X1S = BF;                                // 16-byte pieces / floats per pass-through row
    static_assert(BF % 4 == 0 && BF <= 128 && WM >= 1 && WM * WN == NW && CG * (BM / RUN) <= NTHR, "channel quads, one 32-column tile per wavefront, one depthwise round");
    extern __shared__ __attribute__((aligned(16))) unsigned char sp_smem[];
    const int W = a.W, H = a.H, HW = H * W, tiles = a.tiles, nunits = a.nunits;
    const unsigned win_bytes = (unsigned)(BM + 2 * W + 2) * ROWB;
    const unsigned win_lds = (win_bytes + 15u) & ~15u;
    unsigned char* win = sp_smem;                                        // fp32 window image: flat pixels [m0 - W - 1, m0 + BM + W + 1) x bf
    float* x1s = reinterpret_cast<float*>(sp_smem + win_lds);           // pass-through rows [BM][BF]
    sh16* Ph = reinterpret_cast<sh16*>(sp_smem + win_lds + (unsigned)BM * X1S * 4u);   // operand planes [BM][PS]
    sh16* Pl = Ph + BM * PS;
    int* mtab = reinterpret_cast<int*>(Pl + BM * PS);                   // [BM] nine tap-valid bits per tile row
    float* taps = reinterpret_cast<float*>(mtab + BM);                  // depthwise weights [9][BF] + bias [BF] of the unit in work
    float* biasl = taps + 10 * BF;                                      // [2][2][BF]: (b2, b1n) of the unit in work / of the next item's unit
    int* ctl = reinterpret_cast<int*>(biasl + 4 * BF);                  // [0] next item, [1] its flags were up at the depthwise phase, [2] ... at the first GEMM, [3] scratch
    const unsigned lds_win = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)sp_smem;
    const unsigned lds_x1 = lds_win + win_lds;
    const unsigned lds_taps = lds_x1 + (unsigned)BM * X1S * 4u + 2u * BM * PS * 2u + BM * 4u;
    const unsigned lds_bias = lds_taps + 10u * BF * 4u;

    const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6), l31 = lane & 31, h = lane >> 5;
    const int wm = wave % WM, wn = wave / WM;
    // the CONTROL wavefront (tickets, flag polls): the last one - its columns are >= BF / 2, so its first epilogue stores nothing to global and the
    // wait hipcc places in front of a poll's result (vmcnt(0): it cannot count across the epilogue's branches) finds only the poll itself
    constexpr int CW = NW - 1;
    const int TX = (tiles + 7) >> 3;
    gu32s* const sync = (gu32s*)a.sync;
    gu32s* const flags = sync + STAGE_FLAGS;

    // ---- tickets (the control wavefront) ----------------------------------------------------------------------------------------------------
    // item = unit << 20 | tile, or -1.  Queue q = tiles [q TX, (q+1) TX) of every unit, unit-major.
    int myq = (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u);     // HW_REG_XCC_ID[3:0]: this workgroup's XCD (speed only)
    bool exhausted = false;
    auto queue_n = [&](int q) { const int n = tiles - q * TX; return n < 0 ? 0 : (n > TX ? TX : n); };
    auto decode = [&](int q, int k, int nq) { const int u = k / nq; return (u << 20) | (q * TX + (k - u * nq)); };
    auto fetch_blocking = [&]() -> int {
        for (int tries = 0; tries < 8; ++tries) {
            const int nq = queue_n(myq);
            if (nq > 0) {
                int k = 0;
                if (lane == 0) k = (int)__hip_atomic_fetch_add(sync + myq * STAGE_HEAD_STRIDE, 1u, YN_RLX_AGENT);
                k = __builtin_amdgcn_readfirstlane(k);
                if (k < nunits * nq) return decode(myq, k, nq);
            }
            myq = (myq + 1) & 7;
        }
        exhausted = true;
        return -1;
    };
    int pend = 0;
    bool pend_on = false;
    auto fetch_issue = [&]() {
        pend_on = false;
        if (!exhausted && queue_n(myq) > 0) {
            if (lane == 0) pend = (int)__hip_atomic_fetch_add(sync + myq * STAGE_HEAD_STRIDE, 1u, YN_RLX_AGENT);
            pend_on = true;
        }
    };
    auto fetch_resolve = [&]() -> int {
        if (exhausted) return -1;
        if (pend_on) {
            const int k = __builtin_amdgcn_readfirstlane(pend), nq = queue_n(myq);
            if (k < nunits * nq) return decode(myq, k, nq);
            myq = (myq + 1) & 7;
        }
        return fetch_blocking();
    };
    // ready flags of an item's inputs: lanes 0..2 read (u-1, T-1), (u-1, T), (u-1, T+1) (clamped to the tile range: a duplicate of (u-1, T))
    unsigned pollv = 1u;
    auto poll_issue = [&](int item) {
        const int u = item >> 20, T = item & 0xfffff;
        int tt = T - 1 + (lane < 3 ? lane : 1);
        tt = tt < 0 ? 0 : (tt >= tiles ? tiles - 1 : tt);
        pollv = __hip_atomic_load(flags + (u - 1) * tiles + tt, YN_RLX_AGENT);
    };
    auto poll_ready = [&]() -> bool { return __builtin_amdgcn_ballot_w64(pollv != 0u) == ~0ull; };
    auto wait_ready = [&](int item) {                                   // bounded spin (the control wavefront)
        if ((item >> 20) == 0) return;
        const unsigned long long t0 = wall_clock64();
        for (;;) {
            poll_issue(item);
            if (poll_ready()) return;
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 200000000ull) {                    // 2 s of the 100 MHz counter: something is broken - say so and go on
                if (lane == 0) __hip_atomic_store(sync + STAGE_TIMEOUT, 1u, YN_RLX_AGENT);
                return;
            }
        }
    };

    // ---- DMA issue helpers (addresses from an OPAQUE copy of the thread index: unit_pipe_kernel) ----
    const int t1_lim = ((a.M * (int)ROWB + 15) & ~15) - 16;
    auto issue_window = [&](int item) {
        const int tl = item & 0xfffff;
        const float* t1 = a.u[item >> 20].t1;
        int tt = t;
        asm volatile("" : "+v"(tt));
        const int gs = (tl * BM - W - 1) * (int)ROWB + tt * 16;
        const int nch = (int)((win_bytes + 15u) >> 4);
        for (int c0 = 0; c0 < nch; c0 += NTHR) {
            int src = gs + c0 * 16;
            src = src < 0 ? 0 : (src > t1_lim ? t1_lim : src);
            if (c0 + tt < nch) dma16_sc1(t1, (unsigned)src, lds_win + (unsigned)(c0 + wave * 64) * 16u);
        }
    };
    auto issue_x1 = [&](int item) {
        const int tl = item & 0xfffff;
        const float* x1 = a.u[item >> 20].x1;
        const unsigned x1_ld = (unsigned)a.u[item >> 20].x1_ld;
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
            const unsigned src = (unsigned)m * x1_ld * 4u + (unsigned)j * 16u;
            if (c < nch) dma16_sc1(x1, src, lds_x1 + (unsigned)(c0 + wave * 64) * 16u);
        }
    };
    // depthwise taps + bias -> taps; (b2, b1n) -> biasl[par]: whole 16-byte pieces (BF % 4 == 0), plain loads (weights are read-only)
    auto issue_taps = [&](int unit, int par) {
        const StageUnit& u = a.u[unit];
        int tt = t;
        asm volatile("" : "+v"(tt));
        constexpr int nW = 9 * BF / 4, nB = BF / 4;
#pragma unroll
        for (int c0 = 0; c0 < nW; c0 += NTHR)
            if (c0 + tt < nW) dma16(u.wdw, (unsigned)(c0 + tt) * 16u, lds_taps + (unsigned)(c0 + wave * 64) * 16u);
        if (wave == 1 % NW && lane < nB) dma16(u.bdw, (unsigned)lane * 16u, lds_taps + 9u * BF * 4u);
        if (wave == 2 % NW && lane < nB) dma16(u.b2, (unsigned)lane * 16u, lds_bias + (unsigned)(2 * par) * BF * 4u);
        if (wave == 3 % NW && lane < nB) dma16(u.b1n, (unsigned)lane * 16u, lds_bias + (unsigned)(2 * par + 1) * BF * 4u);
    };
    static_assert(BF / 4 <= 64, "one DMA instruction per bias row");

    // ---- the first item ----
    if (wave == CW) {
        const int c = fetch_blocking();
        if (c >= 0) wait_ready(c);
        if (lane == 0) ctl[3] = c;
    }
    __syncthreads();
    int cur = __builtin_amdgcn_readfirstlane(ctl[3]);
    if (cur >= 0) {
    issue_window(cur);
    issue_x1(cur);
    issue_taps(cur >> 20, 0);

    // ---- both GEMMs' B fragments of this wavefront's columns (unit_pipe_kernel's load_step: no masks, clamped octet / column) ----
    sh16x8 bw2[S][2], bw1[S][2];
    const int ncol = wn * 32 + l31;
    const unsigned lane_w = ((unsigned)h * NPAD + (unsigned)(ncol < NPAD ? ncol : NPAD - 1)) * 16u;   // byte offset of this lane's fragment inside k-step 0 of a pack
    auto step_off = [&](unsigned lw, int s) {
        const unsigned kq = (unsigned)(s * 2 + h < KQ ? s * 2 + h : KQ - 1);
        return (s * 2 + 1 < KQ) ? lw + (unsigned)s * (2u * NPAD * 16u) : lw - (unsigned)h * (NPAD * 16u) + kq * (NPAD * 16u);
    };
    // Everything this wavefront has issued so far - ticket atomics, flag polls, the first item's DMA pieces - is OLDER than the 4 S fragment loads
    // between the two asm statements below (side-effecting asm: the loads cannot be scheduled across either), so the counted wait behind them
    // retires exactly the DMA pieces and leaves the fragments in flight under the first depthwise phase.  Explicit: nothing here relies on
    // a wait the compiler happens to place; tests/test_capi_cpu.py counts the loads between the markers in the ISA.
    asm volatile("; YN_STAGE_FRAG_BEGIN" ::: "memory");
    {
        const StageUnit& u = a.u[cur >> 20];
#pragma unroll
        for (int s = 0; s < S; ++s) {
            bw2[s][0] = *reinterpret_cast<const sh16x8*>(reinterpret_cast<const char*>(u.Ws2h) + step_off(lane_w, s));
            bw2[s][1] = *reinterpret_cast<const sh16x8*>(reinterpret_cast<const char*>(u.Ws2l) + step_off(lane_w, s));
        }
#pragma unroll
        for (int s = 0; s < S; ++s) {
            bw1[s][0] = *reinterpret_cast<const sh16x8*>(reinterpret_cast<const char*>(u.Ws1h) + step_off(lane_w, s));
            bw1[s][1] = *reinterpret_cast<const sh16x8*>(reinterpret_cast<const char*>(u.Ws1l) + step_off(lane_w, s));
        }
    }
    asm volatile("; YN_STAGE_FRAG_END\n\ts_waitcnt vmcnt(%0)" ::"i"(4 * S) : "memory");
    const int cq = t % CG, pl = t / CG;
    const bool worker = pl < BM / RUN;
    {   // K tail of both planes: zero once
        constexpr int padn = PS - BF;
        for (int i = t; i < BM * padn; i += NTHR) { const int r = i / padn, c2 = BF + i - r * padn; Ph[r * PS + c2] = (sh16)0.0f; Pl[r * PS + c2] = (sh16)0.0f; }
    }
    float amax = 0.0f;
    constexpr int jhi = BF >> 1;
    f32x16 acc0, acc1;
    int par = 0;                                                         // which (b2, b1n) pair of biasl belongs to the unit in work

    // refill: k-step s of (rh, rl) replaces the fragments the MFMAs of step s have just read (wave-uniform switch)
    // refill2 (with refill): the OTHER panel `bo` (idle during this GEMM) takes k-step s of (qh, ql) at the same place
    auto gemm = [&](sh16x8 (&bw)[S][2], const void* rh_, const void* rl_, bool refill, sh16x8 (&bo)[S][2], const void* qh_, const void* ql_) {
        const char* rh = reinterpret_cast<const char*>(rh_);
        const char* rl = reinterpret_cast<const char*>(rl_);
        const char* qh = reinterpret_cast<const char*>(qh_);
        const char* ql = reinterpret_cast<const char*>(ql_);
        unsigned lw = lane_w;
        asm volatile("" : "+v"(lw));
#pragma unroll
        for (int k = 0; k < 16; ++k) { acc0[k] = 0.0f; acc1[k] = 0.0f; }
        const sh16* ahp = Ph + (wm * 32 + l31) * PS + h * 8;
        const sh16* alp = Pl + (wm * 32 + l31) * PS + h * 8;
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const sh16x8 ah = *reinterpret_cast<const sh16x8*>(ahp + s * 16);
            const sh16x8 al = *reinterpret_cast<const sh16x8*>(alp + s * 16);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bw[s][0], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bw[s][1], acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bw[s][0], acc1, 0, 0, 0);
            if (refill) {
                const unsigned off = step_off(lw, s);
                bo[s][0] = *reinterpret_cast<const sh16x8*>(qh + off);
                bo[s][1] = *reinterpret_cast<const sh16x8*>(ql + off);
                bw[s][0] = *reinterpret_cast<const sh16x8*>(rh + off);
                bw[s][1] = *reinterpret_cast<const sh16x8*>(rl + off);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = __builtin_fmaf(acc1[r], 1.0f / 2048.0f, acc0[r]);
    };
    auto split2 = [&](int r, int c, float v0, float v1) {
        amax = range_track(range_track(amax, v0), v1);
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 hi, lo;
        hi[0] = (sh16)v0; hi[1] = (sh16)v1;
        lo[0] = (sh16)((v0 - (float)hi[0]) * 2048.0f); lo[1] = (sh16)((v1 - (float)hi[1]) * 2048.0f);
        *reinterpret_cast<h2*>(Ph + r * PS + c) = hi;
        *reinterpret_cast<h2*>(Pl + r * PS + c) = lo;
    };
    auto write_mtab = [&](int item) {
        if (t < BM) {
            const int m0 = (item & 0xfffff) * BM;
            const int rem0 = m0 % HW;
            const int y0 = rem0 / W, x0 = rem0 - y0 * W;
            const int q = x0 + t;
            const int dy = (int)(((float)q + 0.5f) * a.inv_w);
            const int x = q - dy * W;
            const int yy = y0 + dy;
            const int y = yy - (int)(((float)yy + 0.5f) * a.inv_h) * H;
            const int yb = (y >= 1 ? 1 : 0) | 2 | (y + 1 < H ? 4 : 0), xb = (x >= 1 ? 1 : 0) | 2 | (x + 1 < W ? 4 : 0);
            int bits = 0;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
                if ((yb >> ky) & 1) bits |= xb << (3 * ky);
            mtab[t] = m0 + t < a.M ? bits : 0;
        }
    };
    write_mtab(cur);
    auto publish = [&](int item) {                                       // every storing wavefront has drained and passed a barrier
        if (t == 0) __hip_atomic_store(flags + (item >> 20) * tiles + (item & 0xfffff), 1u, YN_RLX_AGENT);
    };

    // One item.  `prev`: the item whose flag is still to be raised (PUB_EARLY = false), or -1.
#ifdef YN_EXP_STAGE_TIMING
    // per-phase cycle sums of wavefront 1 (lane 0 keeps them in LDS behind ctl), printed by a sample of workgroups
    long long* dbg = reinterpret_cast<long long*>(ctl + 16);
    if (t < 24) dbg[t] = 0;
    long long dprev = __builtin_readcyclecounter();
    const long long T_start = dprev;
#define YN_TS(i) do { if (wave == 1 % NW) { const long long now_ = __builtin_readcyclecounter(); if (lane == 0) dbg[i] += now_ - dprev; dprev = now_; } } while (0)
#define YN_CNT(i) do { if (wave == 1 % NW && lane == 0) dbg[i] += 1; } while (0)
#else
#define YN_TS(i)
#define YN_CNT(i)
#endif
    bool pub_out = false;
    auto do_tile = [&](const int cur, const int prev) __attribute__((always_inline)) -> int {
        const int unit = cur >> 20, tile = cur & 0xfffff;
        const StageUnit& U = a.u[unit];
        const int m0 = tile * BM;
        const int nrows = a.M - m0 < BM ? a.M - m0 : BM;
        YN_TS(15);
        lds_barrier();      // (1) window, pass-through rows, taps, biases have landed; mtab is written
        YN_TS(0); YN_CNT(16);

        // ---- depthwise 3x3 from the LDS window -> split planes (unit_pipe_kernel) ----
        if (worker) {
            float4 wd[9];
#pragma unroll
            for (int k = 0; k < 9; ++k) wd[k] = *reinterpret_cast<const float4*>(taps + k * BF + 4 * cq);
            const float4 bd = *reinterpret_cast<const float4*>(taps + 9 * BF + 4 * cq);
            const int r0 = RUN * pl;
            int bits[RUN];
            {
                const int4 b4 = *reinterpret_cast<const int4*>(mtab + r0);
                bits[0] = b4.x; bits[1] = b4.y; bits[2] = b4.z; bits[3] = b4.w;
            }
            const unsigned char* wp = win + ((unsigned)r0 * BF + 4 * cq) * 4u;
            float4 acc[RUN];
#pragma unroll
            for (int i = 0; i < RUN; ++i) acc[i] = bd;
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                float4 row[RUN + 2];
#pragma unroll
                for (int i = 0; i < RUN + 2; ++i) row[i] = *reinterpret_cast<const float4*>(wp + (unsigned)((ky * W + i) * BF) * 4u);
#pragma unroll
                for (int i = 0; i < RUN; ++i)
#pragma unroll
                    for (int kx = 0; kx < 3; ++kx) {
                        const bool ok = (bits[i] >> (ky * 3 + kx)) & 1;
                        float4 v = row[i + kx];
                        v = make_float4(ok ? v.x : 0.0f, ok ? v.y : 0.0f, ok ? v.z : 0.0f, ok ? v.w : 0.0f);
                        vfma(acc[i], v, wd[ky * 3 + kx]);
                    }
            }
#pragma unroll
            for (int i = 0; i < RUN; ++i) {
                float o[4] = {acc[i].x, acc[i].y, acc[i].z, acc[i].w};
                typedef _Float16 hv __attribute__((ext_vector_type(4)));
                hv hi, lo;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    asm volatile("" : "+v"(o[j]));                       // the fp32 value is the result (no v_fma_mixlo_f16: DESIGN 4.1)
                    amax = range_track(amax, o[j]);
                    hi[j] = (sh16)o[j];
                    lo[j] = (sh16)((o[j] - (float)hi[j]) * 2048.0f);
                }
                *reinterpret_cast<hv*>(Ph + (r0 + i) * PS + 4 * cq) = hi;
                *reinterpret_cast<hv*>(Pl + (r0 + i) * PS + 4 * cq) = lo;
            }
        }
        YN_TS(1);
        if (!PUB_EARLY && prev >= 0) vm_drain();                         // the previous item's write-through rows have left this wavefront
        lds_barrier();      // (2) planes complete; the window buffer and the taps are free
        YN_TS(2);
        if (!PUB_EARLY && prev >= 0) publish(prev);
        // The NEXT item is taken HERE, not a tile ahead: a ticket held is a ticket nobody else can start, and an item's inputs are the tickets
        // ~(tiles per unit of a queue) before it - with three tickets per workgroup in hand (first form: this tile, the next one, the one
        // after in flight) five of six tiles found their inputs unfinished and waited ~7 k cycles at their end.
        if (wave == CW) fetch_issue();
        YN_TS(3);
        gemm(bw2, nullptr, nullptr, false, bw1, nullptr, nullptr);
        YN_TS(4);
        lds_barrier();      // (3) every wavefront is done reading the planes (the epilogue writes x2' into them)
        YN_TS(5);
        if (wave == CW) {                                                 // the ticket (behind the barrier: only this wavefront waits for the atomic), and a look at its inputs' flags (answer: under epilogue 1)
            const int n = fetch_resolve();
            if (n >= 0 && (n >> 20) > 0) poll_issue(n); else pollv = 1u;
            if (lane == 0) ctl[0] = n;
        }

        // ---- y = relu(acc + b2): (x1, y) pairs -> global (write-through: the next unit's pass-through half), or split into the planes as x2' ----
        {
            const __amdgpu_buffer_rsrc_t ro = buf_rsrc(U.out);
            const int n = ncol;
            const float bias = n < BF ? biasl[(2 * par) * BF + n] : 0.0f;
            const bool to_global = n < jhi;
            const bool to_plane = n >= jhi && n < BF;
            const float* xr = x1s + (wm * 32 + 4 * h) * X1S + (n < BF ? n : 0);
#pragma unroll
            for (int g8 = 0; g8 < 4; ++g8) {
                if (wm * 32 + 8 * g8 < nrows) {
                    float y[4], xv[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) xv[q] = xr[(q + 8 * g8) * X1S];
#pragma unroll
                    for (int q = 0; q < 4; ++q) y[q] = __builtin_fmaxf(acc0[4 * g8 + q] + bias, 0.0f);
                    if (to_global) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const int row = m0 + wm * 32 + q + 8 * g8 + 4 * h;
                            buf_store_b64<16>(ro, (unsigned)(row * BF + 2 * n) * 4u, make_float2(xv[q], y[q]));
                        }
                    }
                    if (to_plane) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) split2(wm * 32 + q + 8 * g8 + 4 * h, 2 * (n - jhi), xv[q], y[q]);
                    }
                }
            }
        }
        if (wave == CW) {
            const bool r = poll_ready();
            if (lane == 0) ctl[2] = r ? 1 : 0;
        }
        YN_TS(6);
        lds_barrier();      // (4) x2' complete; the pass-through buffer is free; ctl[0] = the next item, ctl[2] = its inputs are there
        YN_TS(7);
        const int nxt = __builtin_amdgcn_readfirstlane(ctl[0]);
        const bool have_next = nxt >= 0;
        const bool rdy = have_next && __builtin_amdgcn_readfirstlane(ctl[2]) != 0;
        const bool sw = have_next && (nxt >> 20) != unit;               // the next item belongs to another unit: its taps, biases and both matrices follow under this tile's second GEMM
        const StageUnit& UN = a.u[have_next ? (nxt >> 20) : unit];
        if (sw) { issue_taps(nxt >> 20, par ^ 1); YN_CNT(20); }
        if (rdy) { issue_window(nxt); write_mtab(nxt); issue_x1(nxt); YN_CNT(18); }
        YN_TS(8);
        // ---- the next unit's pw1 on x2' -> global (write-through: the next unit's depthwise input); both panels are refilled behind its MFMAs (pw2's is idle) ----
        gemm(bw1, UN.Ws1h, UN.Ws1l, sw, bw2, UN.Ws2h, UN.Ws2l);
        YN_TS(9);
        vm_drain();         // the next item's pieces, taps, fragments; this tile's pair stores
        // ... and hipcc is TOLD that the refilled fragments have landed (an empty asm that reads and writes the panels: it places its own wait for
        // them here, behind the drain, where it is free).  Without it the wait sits at the next tile's GEMMs with the count of the loads hipcc knows
        // of - vmcnt(1), vmcnt(0) - which retires whatever DMA pieces are in flight at that point.
        if constexpr (S == 8) {
            asm volatile("" : "+v"(bw2[0][0]), "+v"(bw2[0][1]), "+v"(bw2[1][0]), "+v"(bw2[1][1]), "+v"(bw2[2][0]), "+v"(bw2[2][1]), "+v"(bw2[3][0]), "+v"(bw2[3][1]),
                              "+v"(bw2[4][0]), "+v"(bw2[4][1]), "+v"(bw2[5][0]), "+v"(bw2[5][1]), "+v"(bw2[6][0]), "+v"(bw2[6][1]), "+v"(bw2[7][0]), "+v"(bw2[7][1]));
            asm volatile("" : "+v"(bw1[0][0]), "+v"(bw1[0][1]), "+v"(bw1[1][0]), "+v"(bw1[1][1]), "+v"(bw1[2][0]), "+v"(bw1[2][1]), "+v"(bw1[3][0]), "+v"(bw1[3][1]),
                              "+v"(bw1[4][0]), "+v"(bw1[4][1]), "+v"(bw1[5][0]), "+v"(bw1[5][1]), "+v"(bw1[6][0]), "+v"(bw1[6][1]), "+v"(bw1[7][0]), "+v"(bw1[7][1]));
        } else {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                asm volatile("" : "+v"(bw2[s][0]), "+v"(bw2[s][1]), "+v"(bw1[s][0]), "+v"(bw1[s][1]));
            }
        }
        YN_TS(10);
        {
            const __amdgpu_buffer_rsrc_t rt = buf_rsrc(U.t1n);
            const float bias = ncol < BF ? biasl[(2 * par + 1) * BF + ncol] : 0.0f;
            // gemm_epilogue_impl's 16-byte form: an accumulator quad transposed inside its four lanes, lane j stores row j x 4 columns
            const int j = lane & 3;
            const int nq = wn * 32 + (l31 & ~3);
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float v0 = __builtin_fmaxf(acc0[4 * g + 0] + bias, 0.0f), v1 = __builtin_fmaxf(acc0[4 * g + 1] + bias, 0.0f);
                float v2 = __builtin_fmaxf(acc0[4 * g + 2] + bias, 0.0f), v3 = __builtin_fmaxf(acc0[4 * g + 3] + bias, 0.0f);
                {
                    const float s0 = (j & 1) ? v0 : v1, s1 = (j & 1) ? v2 : v3;
                    const float r0 = quad_xor1(s0), r1 = quad_xor1(s1);
                    if (j & 1) { v0 = r0; v2 = r1; } else { v1 = r0; v3 = r1; }
                }
                {
                    const float s0 = (j & 2) ? v0 : v2, s1 = (j & 2) ? v1 : v3;
                    const float r0 = quad_xor2(s0), r1 = quad_xor2(s1);
                    if (j & 2) { v0 = r0; v1 = r1; } else { v2 = r0; v3 = r1; }
                }
                const int m = m0 + wm * 32 + 8 * g + 4 * h + j;
                if (m < m0 + nrows && nq < BF) buf_store_b128<16>(rt, (unsigned)(m * BF + nq) * 4u, make_float4(v0, v1, v2, v3));
            }
        }
        YN_TS(11);
        // A workgroup never WAITS while it sits on an unpublished tile (two workgroups holding (u, T+1) -> (u+1, T) and (u, T) -> (u+1, T+1)
        // would wait for each other's flag): the deferred form publishes before the bounded wait too
        const bool block = have_next && !rdy;
        pub_out = PUB_EARLY || !have_next || block;
        if (pub_out) {
            vm_drain();
            lds_barrier();
            publish(cur);
        }
        YN_TS(12);
        if (block) {
            YN_CNT(19);
            if (wave == CW) wait_ready(nxt);
            lds_barrier();
            issue_window(nxt); write_mtab(nxt); issue_x1(nxt);
            vm_drain();
        }
        YN_TS(13);
        par ^= sw ? 1 : 0;
        return nxt;
    };
    {
        int nxt = do_tile(cur, -1);
        while (nxt >= 0) {
            const int prev = pub_out ? -1 : cur;
            cur = nxt;
            nxt = do_tile(cur, prev);
        }
    }
    range_report(a.ovf, amax);
#ifdef YN_EXP_STAGE_TIMING
    __syncthreads();
    if (t == 64 % NTHR && (blockIdx.x % 41) == 7)
        printf("stagewg blk %d xcc %d tiles %lld rdy %lld block %lld sw %lld | total %lld | bar1 %lld dw %lld bar2 %lld issue %lld gemm1 %lld bar3 %lld epi1 %lld bar4 %lld x1 %lld gemm2 %lld drain %lld epi2 %lld pub %lld blockwait %lld top %lld\n",
               (int)blockIdx.x, (int)(__builtin_amdgcn_s_getreg(20 | (3 << 11)) & 7u), dbg[16], dbg[18], dbg[19], dbg[20], __builtin_readcyclecounter() - T_start,
               dbg[0], dbg[1], dbg[2], dbg[3], dbg[4], dbg[5], dbg[6], dbg[7], dbg[8], dbg[9], dbg[10], dbg[11], dbg[12], dbg[13], dbg[15]);
#endif
    }   // cur >= 0

    // ---- leave: the last workgroup out zeroes the polled words for the next launch ----
    vm_drain();
    __syncthreads();
    if (t == 0) {
        const unsigned old = __hip_atomic_fetch_add(sync + STAGE_EXIT, 1u, YN_RLX_AGENT);
        ctl[3] = old == gridDim.x - 1 ? 1 : 0;
    }
    __syncthreads();
    if (ctl[3]) {
        for (int i = t; i < nunits * tiles; i += NTHR) __hip_atomic_store(flags + i, 0u, YN_RLX_AGENT);
        if (t < 8) __hip_atomic_store(sync + t * STAGE_HEAD_STRIDE, 0u, YN_RLX_AGENT);
        if (t == 8) __hip_atomic_store(sync + STAGE_EXIT, 0u, YN_RLX_AGENT);
    }
}

static size_t stage_pipe_lds(int bf, int W, int BM)
{
    const size_t win = (((size_t)(BM + 2 * W + 2) * bf * 4 + 15) & ~(size_t)15);
    return win + (size_t)BM * bf * 4 + (size_t)2 * BM * plane_stride(bf) * 2 + (size_t)BM * 4 + (size_t)14 * bf * 4 + 64 + 192;      // (+ YN_EXP_STAGE_TIMING's phase sums)
}

size_t stage_sync_words(int tiles, int nunits) { return (size_t)STAGE_FLAGS + (size_t)tiles * nunits; }

// true = launched (or, dry: would be).  false = no form for this shape or fewer than min_tiles tiles (the caller launches the units one by one).
// The tile count of the form taken goes into the arguments here (a.tiles is ignored on entry).
bool launch_stage_pipe(StageArgs a, int bf, int pub_early, int min_tiles, size_t sync_bytes, hipStream_t s, bool dry)
{
    if (a.nunits < 2 || a.nunits > YN_STAGE_MAX || (a.M & 7)) return false;
    if ((double)a.M * bf * 8.0 >= 4.0e9) return false;                    // 32-bit byte offsets (DMA pieces, raw-buffer stores)
    // workgroups: tickets in hand (~1.65 per workgroup) against the tiles of one unit decide how often an item finds its inputs unfinished
    static const int wg_env = getenv("YN_STAGE_G") ? atoi(getenv("YN_STAGE_G")) : 0;
    // 416 x 416 / bs 32, stage 3: 384 / 448 / 512 workgroups all run 96-98 us (fewer stalls against fewer slots); the narrow branches (<= 48
    // channels: 52 KB of LDS, 168 registers) fit three workgroups per CU: 0.5x bs 128 stage 2 149 -> 130 us with 768
    // Under contention (bench.py's four handles on four streams, same box, 4 x 200-step runs): per-unit launches 47.62 k images/s, this kernel with
    // 320 / 384 / 448 / 512 workgroups 47.51 / 47.43 / 47.32 / 47.22 k - a workgroup that waits for a flag holds a CU slot another stream's kernel
    // could use, and the per-unit launches' idle tails are filled by the other streams anyway.  One stream: 384 ... 512 all +0.4 %.  384.
    const int wg4 = wg_env > 0 ? wg_env : (bf <= 48 ? 768 : 384);
    // two four-wavefront workgroups per CU where their windows fit (80 KB each), else one of eight (twice the rows per tile; 608 x 608 stage 3: W = 38)
#define YN_SP(BFv, NWv)                                                                                                  \
    if (bf == BFv) {                                                                                                     \
        constexpr int BM = 32 * (NWv / (BFv <= 64 ? 2 : 4));                                                             \
        constexpr size_t LDS_MAX = (size_t)(NWv == 4 ? 80 : 160) * 1024;                                                 \
        const size_t lds = stage_pipe_lds(bf, a.W, BM);                                                                  \
        a.tiles = (a.M + BM - 1) / BM;                                                                                   \
        if (lds <= LDS_MAX) {                                                                                            \
            if (a.tiles < min_tiles || a.tiles >= (1 << 20) || stage_sync_words(a.tiles, a.nunits) * sizeof(unsigned) > sync_bytes) return false; \
            if (dry) return true;                                                                                        \
            unsigned g = (unsigned)(NWv == 4 ? wg4 : 256);                                                               \
            if (g > (unsigned)(a.tiles * a.nunits)) g = (unsigned)(a.tiles * a.nunits);                                  \
            if (pub_early) {                                                                                             \
                static unsigned long long attr = 0;                                                                      \
                if (attr_pending(attr)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stage_pipe_kernel<BFv, NWv, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX); \
                set_last_kernel_name("stage_pipe_kernel<" #BFv "," #NWv ",true>");                                       \
                hipLaunchKernelGGL((stage_pipe_kernel<BFv, NWv, true>), dim3(g), dim3(64 * NWv), lds, s, a);             \
            } else {                                                                                                     \
                static unsigned long long attr = 0;                                                                      \
                if (attr_pending(attr)) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(stage_pipe_kernel<BFv, NWv, false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_MAX); \
                set_last_kernel_name("stage_pipe_kernel<" #BFv "," #NWv ",false>");                                      \
                hipLaunchKernelGGL((stage_pipe_kernel<BFv, NWv, false>), dim3(g), dim3(64 * NWv), lds, s, a);            \
            }                                                                                                            \
            return true;                                                                                                 \
        }                                                                                                                \
    }
    YN_SP(116, 4) YN_SP(116, 8)
    YN_SP(96, 4) YN_SP(96, 8)
    YN_SP(48, 4) YN_SP(48, 8)
    YN_SP(24, 4) YN_SP(24, 8)
#undef YN_SP
    return false;
}

}  // namespace ynk

// The persistent recurrent-layer kernel (one launch = one layer, both directions, all time steps).
//
// Decomposition: a workgroup = 4 waves, one wave per SIMD, each wave owns 32 sequences for the
// whole sequence length: its cell state c (fp32, parked in AGPRs) and its hidden state h (bf16
// B-fragments, VGPRs) never leave the register file, so there is no cross-wave exchange of state
// and no grid sync.  Per time step the gate pre-activations
//     gates^T[4H x 32] = [W_hh | W_ih] (A operand, bf16) x [h_{t-1} ; x_t]^T (B operand, bf16)
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The weights do not fit one CU
// (1.5 MiB per direction at H = 256), so they stream L2 -> LDS through a ring of four 8 KiB
// stages by LDS-DMA (global_load_lds_dwordx4), shared by the four waves; forward and backward
// directions run in different workgroups (even / odd blockIdx, i.e. on different XCDs, so each
// XCD's L2 holds one direction's weights).  The layer's input fragments of a step (32 KiB per wave
// above the first layer) are needed once per 32-unit block, i.e. NU times per step: they are
// brought into a private LDS region ONCE per step (also by LDS-DMA, one step ahead, fragment by
// fragment as the last block retires them) instead of being re-streamed from L2 -- re-streaming
// made the per-XCD working set (inputs 4 MiB + weights 1.5 MiB per step) larger than the 4 MiB L2
// and 37 % of all L2 requests missed.
//
// Instruction schedule (all of it fixed at compile time, TdSched below).  One "slot" = one weight
// fragment = one MFMA (two for the first layer's hi/lo products).  Around each MFMA the slot carries
//   * the LDS read of the weight fragment APD slots ahead (register ring),
//   * at most one LDS-DMA piece of the weight stage three stages ahead (a wave moves 2 pieces per
//     stage, at positions 1 and 5, so the texture path never sees a burst and a piece has 24 MFMAs
//     to land),
//   * in the last block, the LDS-DMA piece that replaces a retired input fragment by the next step's,
//   * a few instructions of the gate nonlinearities of the PREVIOUS 32-unit block (two accumulator
//     sets alternate), so the exp/rcp work runs in the shadow of the matrix pipe.
// At a stage boundary a wave waits only for its own DMA pieces of that stage: s_waitcnt vmcnt(N)
// with N = the number of younger vector-memory operations, counted from the same schedule tables
// that emit them.
#pragma once
#include "common.h"
#include "rnn_layout.h"
#include "rnn_params.h"

// diagnostics (tools/kbench only; profiles/round3_layer_ablation.md): -DTD_ABLATE_MFMA drops the matrix instructions,
// -DTD_ABLATE_GATES the nonlinearities, -DTD_ABLATE_DMA the LDS-DMA pieces, -DTD_ABLATE_LDSREAD the LDS reads of the weight
// fragments, -DTD_ABLATE_WAIT / -DTD_ABLATE_BARRIER the stage boundary's wait / barrier, -DTD_ABLATE_MFMA16 issues the
// matrix work as 16x16x32.  None of them computes the layer; none is defined in the library build.
#ifdef TD_ABLATE_MFMA
#define TD_MFMA(a, b, c) (c)
#elif defined(TD_ABLATE_MFMA16)
// diagnostic (tools/kbench; results are NOT the layer's): the same operand traffic issued as v_mfma_f32_16x16x32_bf16 -- one
// A fragment from LDS feeds two matrix instructions of 16 x 16 x 32 (what a 16-unit x 32-k weight fragment against two
// 16-sequence tiles would do), each accumulating into its own quarter of the slot's accumulator; `f` is the slot index
// of the enclosing schedule.  Same FLOPs, LDS bytes, DMA and gate math per step as the shipped kernel.
template <int P>
__device__ __forceinline__ f32x16 td_mfma16_pair(bf16x8 a, bf16x8 b, f32x16 c) {
    f32x4 q0 = {c[8 * P], c[8 * P + 1], c[8 * P + 2], c[8 * P + 3]};
    f32x4 q1 = {c[8 * P + 4], c[8 * P + 5], c[8 * P + 6], c[8 * P + 7]};
    q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, q0, 0, 0, 0);
    q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b, a, q1, 0, 0, 0);
#pragma unroll
    for (int i = 0; i < 4; ++i) { c[8 * P + i] = q0[i]; c[8 * P + 4 + i] = q1[i]; }
    return c;
}
#define TD_MFMA(a, b, c) td_mfma16_pair<(f & 1)>(a, b, c)
#else
#define TD_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0)
#endif

#include <type_traits>

// gate-math micro-ops a slot may carry, in relative units: a slot that also issues vector-memory
// instructions (LDS-DMA piece, input-fragment load) takes fewer than a bare MFMA + LDS-read slot
#ifndef TD_CAP_FREE
#define TD_CAP_FREE 3
#endif
#ifndef TD_CAP_VMEM
#define TD_CAP_VMEM 1
#endif

template <int N, class F>
__device__ __forceinline__ void td_static_for(F &&f) {
    if constexpr (N > 0) {
        td_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// The recurrent state (128 fp32 registers per lane at H = 256) is only touched once per time step,
// so it is parked in the accumulator half of the register file between uses: the VALU-addressable
// half stays free for the operand prefetch rings.
__device__ __forceinline__ float td_park(float v) {
    float a;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
    return a;
}
__device__ __forceinline__ float td_unpark(float a) {
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

#define TD_RING_FRAGS 8   // weight fragments per ring stage
#define TD_RING_DEPTH 4   // ring stages
#define TD_RING_BYTES (TD_RING_FRAGS * TD_FRAG_BYTES)

// HEAD (the sampler's top layer): the step's weight stream ends with 2 KH more fragments, the hi and lo parts of
// fc_w * ln_w * (1 + FiLM scale) per 16 hidden units, and the kernel multiplies them with the h_t fragments it has just
// produced: it leaves per (step, batch block, direction) the partial projection and sum h, sum h^2 instead of the
// activations (rnn.hip: td_head_fold_kernel, td_head_combine_kernel).
template <int NU, int KX, bool L0, bool GRU, bool HEAD = false>
struct TdLayerTraits {
    static constexpr int NG = GRU ? 3 : 4;
    static constexpr int KH = 2 * NU;
    static constexpr int XFR = L0 ? 2 * KX : KX;
    static constexpr int FU = 4 + (KH + XFR) * NG;  // 4 bias fragments lead each unit block
    static constexpr int NFB = NU * FU;             // fragments of the unit blocks
    static constexpr int HF = HEAD ? 2 * KH : 0;    // head fragments behind them
    static constexpr int NF = NFB + HF;             // weight fragments per time step
    static constexpr int NS = (NF + TD_RING_FRAGS - 1) / TD_RING_FRAGS;  // ring stages per time step
    static constexpr int LDS_W = TD_RING_DEPTH * TD_RING_BYTES;
    static constexpr int LDS_X = L0 ? 0 : TD_BLOCK_WAVES * KX * TD_FRAG_BYTES;  // this step's input fragments
    static constexpr int LDS_TOTAL = LDS_W + LDS_X;
    static constexpr int APD = 8;                                   // weight fragments read ahead of their MFMA
    static constexpr int PIECES = TD_RING_FRAGS / TD_BLOCK_WAVES;  // LDS-DMA pieces per wave and stage
    static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
};

// ---------------------------------------------------------------------------------------------
// compile-time instruction schedule of one time step
// ---------------------------------------------------------------------------------------------
template <int NU, int KX, bool L0, bool GRU, bool SAVE, bool HEAD = false>
struct TdSched {
    using T = TdLayerTraits<NU, KX, L0, GRU, HEAD>;
    static constexpr int NG = T::NG, KH = T::KH, FU = T::FU, NF = T::NF, NFB = T::NFB, HF = T::HF, NS = T::NS, APD = T::APD;
    static_assert(!(HEAD && SAVE) && !(HEAD && L0), "the fused head is an inference variant of the layers above the first");
    static constexpr int NOPS = GRU ? 7 : 12;  // gate-math micro-ops per accumulator row
    static constexpr int NQ = 16 * NOPS;       // per 32-unit block
    // Pieces issued while stage j is read belong to stage j + LEAD and land in the buffer stage j - 2 was read
    // from.  Every read of stage j - 2 has been consumed by an MFMA issued before stage j opens (the register
    // ring holds 8 = one stage of fragments), so no wave needs to drain its LDS reads at the boundary.
    static constexpr int LEAD = TD_RING_DEPTH - 2;
    static constexpr int PIECE_STEP = 2;  // pieces go out at positions 1, 3 (both in one slot, back to back: +1.7 cycles per fragment)
    static_assert(T::APD <= TD_RING_FRAGS, "the LDS read ring must not reach back two stages");

    // weight fragment whose LDS read is issued in slot F (-1: none)
    static constexpr int refill_of(int F) { return F + APD < NF ? F + APD : -1; }
    static constexpr int stage_len(int j) { return j == NS - 1 ? NF - TD_RING_FRAGS * (NS - 1) : TD_RING_FRAGS; }
    // piece i goes out with the read of the fragment at position min(2i+1, len-1) of the stage being read
    static constexpr int piece_pos(int j, int i) {
        const int len = stage_len(j);
        return PIECE_STEP * i + 1 < len - 1 ? PIECE_STEP * i + 1 : len - 1;
    }
    static constexpr int wpieces(int F) {  // weight pieces issued in slot F
        const int R = refill_of(F);
        if (R < 0) return 0;
        int n = 0;
        for (int i = 0; i < T::PIECES; ++i) n += piece_pos(R / TD_RING_FRAGS, i) == R % TD_RING_FRAGS;
        return n;
    }
    // input-fragment index whose LDS copy is replaced by the next step's in slot F (-1: none): the last block
    // retires fragment kx with its last gate's MFMA
    static constexpr int xdma_of(int F) {
        if (L0 || F / FU != NU - 1) return -1;
        const int x = F % FU - 4 - KH * NG;
        if (x < 0) return -1;
        return x % NG == NG - 1 ? x / NG : -1;
    }
    // training kernels: vector stores issued by micro-op q (the gate values of 4 accumulator rows go out
    // together, 5 quantities x 16 B per lane, with the op that completes the second row pair)
    static constexpr int save_stores(int q) {
        const int pair = q / (2 * NOPS), w = q % (2 * NOPS);
        return (SAVE && (pair & 1) && (w & 1) && w / 2 == (GRU ? 6 : 8)) ? 5 : 0;
    }
    static constexpr bool boundary(int F) { return refill_of(F) >= 0 && refill_of(F) % TD_RING_FRAGS == 0; }
    // room for gate-math micro-ops beside slot F's MFMA (block 0 of a step has no previous block to finish)
    static constexpr int capacity(int F) {
        if (F / FU == 0) return 0;
        // head slots: the last four multiply the last block's own h_t, i.e. follow every one of its gate micro-ops
        if (F >= NFB && F - NFB >= HF - 4) return 0;
        if (boundary(F)) return 0;  // wait + barrier
        if (wpieces(F) > 0 || xdma_of(F) >= 0) return TD_CAP_VMEM;
        return TD_CAP_FREE;
    }
    struct Tab {
        short qbeg[NF], qend[NF];  // micro-ops [qbeg, qend) of the previous block run in slot F
        short waitn[NS];           // vmcnt operand of the wait that opens stage j
    };
    static constexpr Tab make() {
        Tab t{};
        for (int u = 0; u < NU; ++u) {
            int total = 0;
            for (int f = 0; f < FU; ++f) total += capacity(u * FU + f);
            int cum = 0;
            for (int f = 0; f < FU; ++f) {
                const int F = u * FU + f;
                t.qbeg[F] = total ? (short)((long)cum * NQ / total) : 0;
                cum += capacity(F);
                t.qend[F] = total ? (short)((long)cum * NQ / total) : 0;
            }
        }
        if (HF > 0) {   // the head slots carry the last block's gate micro-ops
            int total = 0;
            for (int f = 0; f < HF; ++f) total += capacity(NFB + f);
            int cum = 0;
            for (int f = 0; f < HF; ++f) {
                const int F = NFB + f;
                t.qbeg[F] = total ? (short)((long)cum * NQ / total) : (short)NQ;
                cum += capacity(F);
                t.qend[F] = total ? (short)((long)cum * NQ / total) : (short)NQ;
            }
        }
        // Stage j opens in slot Fb = 8j - APD.  Its two pieces went out while stage j - LEAD (a full stage) was
        // read, the second one in slot Fl = 8(j - LEAD) + 3 - APD.  Everything issued after that piece and
        // before the wait is younger than both: inside slot Fl the input-fragment piece and the output stores
        // (they follow the weight piece), then all vector-memory instructions of the slots in between.  Where the
        // pieces came from the previous step or the prologue, wait for everything.
        for (int j = 0; j < NS; ++j) {
            int n = 0;
            const int Fb = TD_RING_FRAGS * j - APD, Fl = TD_RING_FRAGS * (j - LEAD) + PIECE_STEP + 1 - APD;
#ifndef TD_WAIT_ALL
            if (j >= LEAD && Fl >= 0) {
                for (int F = Fl; F < Fb; ++F) {
                    int st = (!HEAD && t.qend[F] == NQ && t.qbeg[F] < NQ) ? 2 : 0;   // the block's two output fragments
                    for (int q = t.qbeg[F]; q < t.qend[F]; ++q) st += save_stores(q);
                    n += (F > Fl ? wpieces(F) : 0) + (xdma_of(F) >= 0 ? 1 : 0) + st;
                }
            }
#endif
            t.waitn[j] = (short)(n > 60 ? 60 : n);
        }
        return t;
    }
    static constexpr Tab tab = make();
};

template <int NU, int KX, bool L0, bool GRU, bool SAVE = false, bool HEAD = false>
__global__ __launch_bounds__(256, 1) void td_rnn_layer_kernel(TdLayerParams p) {
    using T = TdLayerTraits<NU, KX, L0, GRU, HEAD>;
    using S = TdSched<NU, KX, L0, GRU, SAVE, HEAD>;
    constexpr int NG = T::NG, KH = T::KH, XFR = T::XFR, FU = T::FU, NS = T::NS, NF = T::NF, NFB = T::NFB, HF = T::HF, APD = T::APD;
    constexpr int NOPS = S::NOPS, NQ = S::NQ, LEAD = S::LEAD;
    __shared__ __attribute__((aligned(16))) char lds[T::LDS_TOTAL];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int dir = p.dirs == 2 ? (blockIdx.x & 1) : 0;
    const int tile = p.dirs == 2 ? (blockIdx.x >> 1) : blockIdx.x;
    const int bblk = tile * TD_BLOCK_WAVES + wave;
    // a wave past the end of the batch repeats the last block's work (same inputs, same outputs, same stores):
    // every wave then issues the same vector-memory stream, which the counted waits below rely on
    const int bb = bblk < p.nbb ? bblk : p.nbb - 1;
#ifdef TD_STAMP
    const uint64_t stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // LDS-DMA pieces (1 KiB = one fragment per wave-instruction; scalar base + 32-bit lane offset).
    // The piece is issued from inline asm: the compiler models an LDS-DMA load as a flat access to both LDS and
    // memory and then drains lgkmcnt AND vmcnt to zero at the next dependent instruction, which would stall the
    // LDS read ring after every piece.  Written this way it does not know about the pieces; they are covered by
    // the counted waits at the stage boundaries and the full wait at the top of every step.  M0 (the LDS base of
    // the DMA) is used by nothing else in this kernel.
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    auto dma = [&](const char *src, unsigned dst) {
#ifdef TD_ABLATE_DMA
        // diagnostic (tools/kbench): no LDS-DMA at all, the MFMAs read whatever the LDS holds -- what issuing the pieces costs
        return;
#endif
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(lane16), "s"(src) : "memory");
    };
    // weights: ring stage j (any step) = fragments 8j .. 8j+7 of the packed stream, wave w moves 2w, 2w+1
    const char *wwave = p.wpacked + (size_t)dir * p.dir_stride + (size_t)wave * T::PIECES * TD_FRAG_BYTES;
    const unsigned ldsw = lds0 + wave * T::PIECES * TD_FRAG_BYTES;
    auto issue_piece = [&](const char *wsrc, int stage, int piece, int buf) {
        const char *sp = wsrc + (size_t)stage * TD_RING_BYTES + piece * TD_FRAG_BYTES;
        asm volatile("" : "+s"(sp));
        dma(sp, ldsw + buf * TD_RING_BYTES + piece * TD_FRAG_BYTES);
    };
    // inputs: this wave's private region, fragment kx at xlds + kx KiB
    char *xreg = lds + T::LDS_W + wave * (KX * TD_FRAG_BYTES);
    const unsigned ldsx = lds0 + T::LDS_W + wave * (KX * TD_FRAG_BYTES);
    auto xrow = [&](int l) { return p.xin + ((size_t)l * p.nbb + bb) * p.in_frags * TD_FRAG_BYTES; };

    float state[NU][16];  // LSTM: cell state c; GRU: hidden state h in fp32 (parked, see td_park)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) state[u][r] = td_park(0.f);

    // prologue: the first LEAD weight stages and the first step's inputs
    td_static_for<LEAD>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int i = 0; i < T::PIECES; ++i) issue_piece(wwave, j % NS, i, j % TD_RING_DEPTH);
    });
    if constexpr (!L0) {
        const char *x0 = xrow(dir ? p.L - 1 : 0);
#pragma unroll
        for (int kx = 0; kx < KX; ++kx) dma(x0 + kx * TD_FRAG_BYTES, ldsx + kx * TD_FRAG_BYTES);
    }
    int ring = TD_RING_DEPTH - 1;  // buffer of the stage being read; advanced at every stage boundary

    // B operand of the bias k-step: ones at k = 0, 1 (the bias rides in as bf16 hi + lo columns)
    bf16x8 ones = {};
    if (lane < 32) { ones[0] = (__bf16)1.0f; ones[1] = (__bf16)1.0f; }

    f32x16 acc[2][4];  // two accumulator sets: block u accumulates into acc[u & 1] while block u-1's gates are finished
    bf16x8 hold[KH];   // h_{t-1}: B operand of the recurrent k-steps
    bf16x8 hn[KH];     // h_t, becomes hold once every block has consumed h_{t-1}
    float T0[2][6];    // gate-math temporaries of the two rows in flight
    f32x4 sbuf[5];     // training: saved gate values of four accumulator rows, stored together
    float hprev = 0.f;
    {
        bf16x8 zero = {};
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) hold[ks] = zero;
    }

    for (int s = 0; s < p.L; ++s) {
        const int l = dir ? p.L - 1 - s : s;
        const bool last_step = s + 1 == p.L;
        // next step's inputs (the last step re-reads its own: the piece count per step stays fixed)
        const char *xnext = xrow(last_step ? l : (dir ? l - 1 : l + 1));
        asm volatile("" : "+s"(xnext));
        // layer 0 keeps its few input fragments in registers
        bf16x8 xf[L0 ? XFR : 1];
        if constexpr (L0) {
            const bf16x8 *xp = (const bf16x8 *)xrow(l) + lane;
#pragma unroll
            for (int i = 0; i < XFR; ++i) xf[i] = __builtin_nontemporal_load(xp + i * 64);
        }
        char *op = p.out + (((size_t)l * p.nbb + bb) * p.out_frags + dir * KH) * TD_FRAG_BYTES + lane * 16;
        float *svb = nullptr;
        if constexpr (SAVE) svb = p.save + (size_t)dir * p.save_dir_stride + (((size_t)l * p.nbb + bb) * NU * 5) * 1024 + lane * 16;

        float hs1 = 0.f, hs2 = 0.f;   // HEAD: sum h_t and sum h_t^2 over this lane's units (of the bf16 values the head would read)
        // the per-step base pointer is laundered so that nothing derived from it is loop-invariant:
        // otherwise every stage's DMA base gets hoisted out of the time loop and spilled
        const char *wstep = wwave;
        asm volatile("" : "+s"(wstep));

        // LDS read of weight fragment R of this step (+ what the schedule attaches to it)
        auto refill = [&](auto rc) -> bf16x8 {
            constexpr int R = decltype(rc)::value;
            constexpr int stage = R / TD_RING_FRAGS, pos = R % TD_RING_FRAGS;
            if constexpr (pos == 0) {
                // my pieces of `stage` have landed (younger loads/stores may stay in flight); after the barrier
                // that holds for every wave, and every wave is done with the buffer the next pieces go to
                constexpr int WN = S::tab.waitn[stage];
#ifndef TD_ABLATE_WAIT
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WN) : "memory");
#endif
#ifndef TD_ABLATE_BARRIER
                __builtin_amdgcn_s_barrier();
#endif
                asm volatile("" ::: "memory");
                ring = (ring + 1) & (TD_RING_DEPTH - 1);
            }
#ifdef TD_ABLATE_LDSREAD
            // diagnostic (tools/kbench): the weight fragments are not read from LDS at all (what the LDS reads cost)
            bf16x8 a = ones;
            asm volatile("" : "+v"(a));
#else
            const bf16x8 a = *(const bf16x8 *)(lds + ring * TD_RING_BYTES + pos * TD_FRAG_BYTES + lane * 16);
#endif
            td_static_for<T::PIECES>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (S::piece_pos(stage, i) == pos) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_piece(wstep, (stage + LEAD) % NS, i, (ring + LEAD) & (TD_RING_DEPTH - 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            return a;
        };
        auto xfrag = [&](int kx) -> bf16x8 { return *(const bf16x8 *)(xreg + kx * TD_FRAG_BYTES + lane * 16); };

        // micro-op q of block ub's gate nonlinearities; rows go in pairs so that the dependent exp -> rcp -> fma
        // chain of one row is separated by the other row's instructions.
        // row r of the accumulator = unit 32u + (r&3) + 8(r>>2) + 4hf
        auto gate_op = [&](auto ubc, auto qc) {
            constexpr int ub = decltype(ubc)::value, q = decltype(qc)::value;
            constexpr int pair = q / (2 * NOPS), w = q % (2 * NOPS), o = w / 2, r = 2 * pair + (w & 1);
            constexpr float P2L2E = TD_GATE_SCALE_TANH;
            f32x16(&a)[4] = acc[ub & 1];
            float(&t)[6] = T0[r & 1];
            float hval = 0.f;
#ifdef TD_ABLATE_GATES
            // diagnostic (tools/kbench): no nonlinearities, h = the first gate's pre-activation -- what the MFMA / LDS / DMA side costs alone
            if constexpr (o == NOPS - 1) hval = a[0][r];
            if constexpr (false) {
#else
            if constexpr (!GRU) {
#endif
                // the accumulators arrive pre-multiplied by -log2(e) (i, f, o) / 2 log2(e) (g): td_pack_weights_kernel
                if constexpr (o == 0) t[0] = __builtin_amdgcn_exp2f(a[0][r]);
                if constexpr (o == 1) t[1] = __builtin_amdgcn_exp2f(a[1][r]);
                if constexpr (o == 2) t[2] = __builtin_amdgcn_exp2f(a[2][r]);
                if constexpr (o == 3) t[3] = __builtin_amdgcn_exp2f(a[3][r]);
                if constexpr (o == 4) t[0] = __builtin_amdgcn_rcpf(1.0f + t[0]);  // i
                if constexpr (o == 5) t[1] = __builtin_amdgcn_rcpf(1.0f + t[1]);  // f
                if constexpr (o == 6) t[2] = __builtin_amdgcn_rcpf(t[2] + 1.0f);  // g = 1 - 2 t[2]
                if constexpr (o == 7) t[3] = __builtin_amdgcn_rcpf(1.0f + t[3]);  // o
                if constexpr (o == 8) {
                    // explicit fma: every instantiation (inference / training) rounds the same way
                    const float gg = __builtin_fmaf(-2.0f, t[2], 1.0f);
                    const float cv = __builtin_fmaf(t[1], td_unpark(state[ub][r]), t[0] * gg);
                    state[ub][r] = td_park(cv);
                    t[4] = cv;
                    if constexpr (SAVE) {
                        sbuf[0][r & 3] = t[0]; sbuf[1][r & 3] = t[1]; sbuf[2][r & 3] = gg; sbuf[3][r & 3] = t[3]; sbuf[4][r & 3] = cv;
                    }
                }
                if constexpr (o == 9) t[5] = __builtin_amdgcn_exp2f(P2L2E * t[4]);
                if constexpr (o == 10) t[5] = __builtin_amdgcn_rcpf(t[5] + 1.0f);
                if constexpr (o == 11) hval = t[3] * __builtin_fmaf(-2.0f, t[5], 1.0f);
            } else {
                // pre-multiplied accumulators: -log2(e) (r, z), 2 log2(e) (both halves of n)
                if constexpr (o == 0) t[0] = __builtin_amdgcn_exp2f(a[0][r]);
                if constexpr (o == 1) t[1] = __builtin_amdgcn_exp2f(a[1][r]);
                if constexpr (o == 2) t[0] = __builtin_amdgcn_rcpf(1.0f + t[0]);  // r
                if constexpr (o == 3) t[1] = __builtin_amdgcn_rcpf(1.0f + t[1]);  // z
                if constexpr (o == 4) t[2] = __builtin_amdgcn_exp2f(__builtin_fmaf(t[0], a[3][r], a[2][r]));
                if constexpr (o == 5) t[2] = __builtin_amdgcn_rcpf(t[2] + 1.0f);
                if constexpr (o == 6) {
                    const float ng = __builtin_fmaf(-2.0f, t[2], 1.0f);
                    const float h = __builtin_fmaf(t[1], td_unpark(state[ub][r]), (1.0f - t[1]) * ng);
                    state[ub][r] = td_park(h);
                    if constexpr (SAVE) {
                        sbuf[0][r & 3] = t[0]; sbuf[1][r & 3] = t[1]; sbuf[2][r & 3] = ng; sbuf[3][r & 3] = a[3][r] * (1.0f / P2L2E); sbuf[4][r & 3] = h;
                    }
                    hval = h;
                }
            }
            if constexpr (S::save_stores(q) > 0) {
                // rows r-3 .. r are complete: one 16-byte store per saved quantity
                float *svp = svb + (size_t)ub * 5 * 1024 + (r & ~3);
#pragma unroll
                for (int k = 0; k < 5; ++k) *(f32x4 *)(svp + k * 1024) = sbuf[k];
            }
            if constexpr (o == NOPS - 1) {
                // rows r-1, r are neighbours in the bf16 fragment: convert them as a pair
                if constexpr ((r & 1) == 0) hprev = hval;
                else {
                    const __bf16 b0 = (__bf16)hprev, b1 = (__bf16)hval;
                    hn[2 * ub + (r >> 3)][(r & 7) - 1] = b0;
                    hn[2 * ub + (r >> 3)][r & 7] = b1;
                    if constexpr (HEAD) {
                        const float f0 = (float)b0, f1 = (float)b1;
                        hs1 += f0 + f1;
                        hs2 = __builtin_fmaf(f0, f0, __builtin_fmaf(f1, f1, hs2));
                    }
                }
            }
            if constexpr (q == NQ - 1 && !HEAD) {
                // h_t of this block goes to HBM for the next layer right away (streaming store: it must not push
                // the weights out of this XCD's L2)
                __builtin_nontemporal_store(hn[2 * ub], (bf16x8 *)(op + (2 * ub) * TD_FRAG_BYTES));
                __builtin_nontemporal_store(hn[2 * ub + 1], (bf16x8 *)(op + (2 * ub + 1) * TD_FRAG_BYTES));
            }
        };

        // Weight fragments are read APD ahead of the MFMA that consumes them (LDS latency is ~100+ cycles,
        // an MFMA 32): the ring A[] is statically indexed, so it lives in registers.  The first read of a step
        // opens stage 0 behind a full wait, which also covers this step's input pieces.
        bf16x8 A[APD];
        td_static_for<(APD < NF ? APD : NF)>([&](auto ic) { A[decltype(ic)::value] = refill(ic); });
        bf16x8 xb[2];  // input fragments, read from LDS one k-step ahead

        td_static_for<NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            td_static_for<FU>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                constexpr int F = u * FU + f;
                if constexpr (f == 0) {
#pragma unroll
                    for (int g = 0; g < 4; ++g)
#pragma unroll
                        for (int r = 0; r < 16; ++r) acc[u & 1][g][r] = 0.f;
                }
                if constexpr (!L0 && f == 4 + KH * NG - 2) xb[0] = xfrag(0);
                const bf16x8 a = A[F % APD];
                f32x16(&ac)[4] = acc[u & 1];
                if constexpr (f < 4) {
                    // bias k-step (LSTM: b_ih + b_hh per gate; GRU: r, z, n-input, n-hidden)
                    ac[f] = TD_MFMA(a, ones, ac[f]);
                } else if constexpr (f - 4 < KH * NG) {
                    // recurrent part: h_{t-1} straight from registers
                    constexpr int ks = (f - 4) / NG, g = (f - 4) % NG;
                    constexpr int ai = (GRU && g == 2) ? 3 : g;
                    ac[ai] = TD_MFMA(a, hold[ks], ac[ai]);
                } else if constexpr (L0) {
                    // first layer: W = Whi + Wlo, x = xhi + xlo in bf16; keep the three leading products
                    constexpr int x = f - 4 - KH * NG;
                    constexpr int part = x & 1, kx = (x >> 1) / NG, g = (x >> 1) % NG;
                    if constexpr (part == 0) {
                        ac[g] = TD_MFMA(a, xf[2 * kx + 1], ac[g]);
                        ac[g] = TD_MFMA(a, xf[2 * kx], ac[g]);
                    } else {
                        ac[g] = TD_MFMA(a, xf[2 * kx], ac[g]);
                    }
                } else {
                    constexpr int x = f - 4 - KH * NG;
                    constexpr int kx = x / NG, g = x % NG;
                    if constexpr (g == 0 && kx + 1 < KX) xb[(kx + 1) % 2] = xfrag(kx + 1);
                    ac[g] = TD_MFMA(a, xb[kx % 2], ac[g]);
                }
                if constexpr (S::refill_of(F) >= 0)
                    A[F % APD] = refill(std::integral_constant<int, (S::refill_of(F) >= 0 ? S::refill_of(F) : 0)>{});
                // the last block retires input fragment kx: bring in the next step's
                if constexpr (S::xdma_of(F) >= 0) {
                    constexpr int kx = S::xdma_of(F) >= 0 ? S::xdma_of(F) : 0;
                    __builtin_amdgcn_sched_barrier(0);
                    dma(xnext + kx * TD_FRAG_BYTES, ldsx + kx * TD_FRAG_BYTES);
                    __builtin_amdgcn_sched_barrier(0);
                }
                // gate math of the previous block, in the shadow of this block's MFMAs
                if constexpr (u > 0) {
                    constexpr int q0 = S::tab.qbeg[F], q1 = S::tab.qend[F];
                    td_static_for<q1 - q0>([&](auto jc) {
                        gate_op(std::integral_constant<int, (u > 0 ? u - 1 : 0)>{},
                                std::integral_constant<int, q0 + decltype(jc)::value>{});
                    });
                }
                // pin the slot: left alone, the scheduler sinks every ds_read next to its MFMA and piles the
                // gate math up behind the last MFMA
                __builtin_amdgcn_sched_barrier(0);
            });
        });

        if constexpr (!HEAD) {
            // the last block's gates have no MFMAs left to hide behind
            td_static_for<NQ>([&](auto qc) { gate_op(std::integral_constant<int, NU - 1>{}, qc); });
        } else {
            // head slots: partial projection of h_t, k-step by k-step (hi and lo part of the folded weights); the k-steps of
            // the blocks before the last run beside the last block's gate math, the last block's own four slots after it
            f32x16 &hacc = acc[NU & 1][0];     // the accumulator set the last block does not use
            // a single unit block: every head slot multiplies that block's own h_t, its gate math has nothing to hide behind
            if constexpr (HF <= 4) td_static_for<NQ>([&](auto qc) { gate_op(std::integral_constant<int, NU - 1>{}, qc); });
            td_static_for<HF>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                constexpr int F = NFB + f;
                if constexpr (f == 0) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) hacc[r] = 0.f;
                }
                const bf16x8 a = A[F % APD];
                hacc = TD_MFMA(a, hn[f / 2], hacc);
                if constexpr (S::refill_of(F) >= 0)
                    A[F % APD] = refill(std::integral_constant<int, (S::refill_of(F) >= 0 ? S::refill_of(F) : 0)>{});
                constexpr int q0 = S::tab.qbeg[F], q1 = S::tab.qend[F];
                td_static_for<q1 - q0>([&](auto jc) {
                    gate_op(std::integral_constant<int, NU - 1>{}, std::integral_constant<int, q0 + decltype(jc)::value>{});
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            // accumulator row r of lane (m, hf) = output channel (r&3) + 8(r>>2) + 4hf of sequence m
            float *hp = p.hpart + ((((size_t)l * p.nbb + bb) * p.dirs + dir) * 64 + lane) * p.hp_stride;
            *(f32x4 *)hp = f32x4{hacc[0], hacc[1], hacc[2], hacc[3]};
            *(f32x4 *)(hp + 4) = f32x4{hacc[4], hacc[5], hacc[6], hacc[7]};
            if (p.hp_stride > 12) {   // more than 16 output channels
                *(f32x4 *)(hp + 8) = f32x4{hacc[8], hacc[9], hacc[10], hacc[11]};
                *(f32x4 *)(hp + 12) = f32x4{hacc[12], hacc[13], hacc[14], hacc[15]};
            }
            *(f32x4 *)(hp + p.hp_stride - 4) = f32x4{hs1, hs2, 0.f, 0.f};
        }

        // h_t replaces h_{t-1} only now: every unit block above consumed the old one
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) hold[ks] = hn[ks];
    }
    // no LDS-DMA piece may still be in flight when the workgroup's LDS is handed on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TD_STAMP
    // diagnostic build only: in-kernel clock = d(memtime) / d(memrealtime) x 100 MHz; the stamps go to a buffer nothing reads
    if (tid == 0) {
        unsigned long long *st = (unsigned long long *)p.save + 2 * blockIdx.x;
        st[0] = __builtin_amdgcn_s_memtime() - stamp_c0;
        st[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
    }
#endif
}

// EXPERIMENT, NOT PART OF THE PRODUCT (round 3; timed by layer16_bench.hip in this directory, never linked into the library).
// The batch-block layer kernel rebuilt on v_mfma_f32_16x16x32_bf16.  It was wired into the library once (own weight packing,
// head fold, dispatch) and passed the forced-family parity tests, then measured against rnn_layer.h on the same boxes
// (profiles/round3_layer_ablation.md): the chip holds 2.2 GHz with it instead of 1.85 GHz, but the wave issues two matrix
// instructions per weight fragment (16 of 32 cycles of vector issue instead of 8) and needs 63 cycles per fragment instead
// of 52.5 -- 12.5-13.1 ms per upper layer against 12.0-12.6 ms.  Not adopted; kept so that the measurement can be repeated.
//
// The persistent recurrent-layer kernel of the batch-block family, inference form (one launch = one layer, both
// directions, all time steps), on v_mfma_f32_16x16x32_bf16.
//
// Same decomposition as rnn_layer.h (which stays the SAVING form of the family, for training): a workgroup = 4 waves, one
// per SIMD, a wave owns 32 sequences for all L steps, cell state fp32 parked in AGPRs, h as bf16 B operands in VGPRs; the
// layer's weights stream L2 -> LDS ring (LDS-DMA, 4 stages of 8 KiB shared by the 4 waves) -> registers; a step's input
// fragments are staged in LDS once; the gate math of unit block u - 1 rides in the slots of block u; counted vmcnt waits.
//
// What differs is the matrix instruction.  At the bench shape the chip runs this kernel at its power limit, and the clock
// it holds depends on the MFMA shape (MI355X_MICROARCH.md, DVFS give-back (7)): the same operand traffic issued as 16x16x32
// ran at 2.09 GHz against 1.85 GHz (profiles/round3_layer_ablation.txt).  With 16 x 16 x 32 a wave's 32 sequences are two
// TILES of 16; a weight fragment is 16 rows x 32 k (1 KiB, as before) and feeds two matrix instructions, one per tile.
//
// Layouts (all private to this kernel family; rnn_layout.h describes the 32x32x16 "F" fragments the other kernels use):
//   * B operand "T fragment" (1 KiB): lane (n = lane & 15, kg = lane >> 4), element j = 4 rt + r holds, for sequence
//     16 t + n of the batch block, hidden unit  32 u + 16 (kg & 1) + 4 (kg >> 1) + 8 rt + r  -- which is exactly what the
//     accumulators of the two 16-row tiles rt = 0, 1 of unit block u leave in that lane (MFMA row i = 4 kg + r), so h_t
//     goes accumulator -> bf16 -> next step's operand without a shuffle.  A layer's output is [l][block][dir][u][t].
//   * one v_permlane16_swap per register turns the T pair (t = 0, 1) of a unit block into the F pair (k-steps 2u, 2u + 1)
//     and back: the top layer writes F fragments (head kernel, td_rnn_read_hidden), the first layer reads F (hi, lo)
//     fragments of its 16 input channels and swaps them into  [x_hi(0..7) | x_lo(0..7) | x_hi(8..15) | x_lo(8..15)].
//   * A operand: lane (i = lane & 15, kg = lane >> 4): row i = unit 32 u + 16 ((i >> 2) & 1) + 4 (i >> 3) + 8 rt + (i & 3),
//     8 k values of group kg in the order of the B operand.  Stream per unit block:
//         [bias: rt x 4 accumulators][recurrent: k-step x rt x gate][input: k-step x rt x gate]
//     First layer: W_ih = hi + lo; the fragment [hi | hi | hi | hi] multiplies the swapped input (W_hi x_hi + W_hi x_lo),
//     and W_lo x_hi of the first 16 channels rides in the bias fragments: [W_lo(0..7) | bias hi, lo | W_lo(8..15) | 0]
//     against [x_hi(0..7) | 1, 1 | x_hi(8..15) | 0].
//   * fused head (HEAD = channel tiles of 16): per unit block and channel tile a hi and a lo fragment of the folded
//     projection weights; the partial projections are swapped into the layout td_head_combine_kernel reads.
#pragma once
#include "../../../timediff_amd/csrc/common.h"
#include "../../../timediff_amd/csrc/rnn_layout.h"
#include "../../../timediff_amd/csrc/rnn_params.h"

#include <type_traits>

#ifndef TD16_PIECE_STEP
#define TD16_PIECE_STEP 2
#endif

#ifndef TD16_OUT_TILES
#define TD16_OUT_TILES 1
#endif
typedef unsigned td16_u32x4 __attribute__((__vector_size__(16)));

#ifdef TD_ABLATE_MFMA
#define TD16_MFMA(a, b, c) (c)
#else
#define TD16_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0)
#endif

#ifndef TD16_CAP_FREE
#define TD16_CAP_FREE 3
#endif
#ifndef TD16_CAP_VMEM
#define TD16_CAP_VMEM 1
#endif

#define TD16_RING_FRAGS 8   // weight fragments per ring stage
#define TD16_RING_DEPTH 4   // ring stages
#define TD16_RING_BYTES (TD16_RING_FRAGS * TD_FRAG_BYTES)

template <int N, class F>
__device__ __forceinline__ void td16_static_for(F &&f) {
    if constexpr (N > 0) {
        td16_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

__device__ __forceinline__ float td16_park(float v) {
    float a;
    asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v));
    return a;
}
__device__ __forceinline__ float td16_unpark(float a) {
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}

// rows of 16 lanes: x = {x0, x1, x2, x3}, y = {y0, y1, y2, y3}  ->  x = {x0, y0, x2, y2}, y = {x1, y1, x3, y3}
__device__ __forceinline__ void td16_swap16(bf16x8 &x, bf16x8 &y) {
    td16_u32x4 a = __builtin_bit_cast(td16_u32x4, x), b = __builtin_bit_cast(td16_u32x4, y);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const auto r = __builtin_amdgcn_permlane16_swap(a[i], b[i], false, false);
        a[i] = r[0];
        b[i] = r[1];
    }
    x = __builtin_bit_cast(bf16x8, a);
    y = __builtin_bit_cast(bf16x8, b);
}
// a value per (tile t, lane (n, kg)) as a = tile 0, b = tile 1  ->  the 32x32 accumulator arrangement: lane (m = 16 t + n, hf)
// holds in a what lane group kg = hf held, in b what kg = hf + 2 held
__device__ __forceinline__ void td16_to_rows32(float &a, float &b) {
    unsigned x = __builtin_bit_cast(unsigned, a), y = __builtin_bit_cast(unsigned, b);
    const auto r = __builtin_amdgcn_permlane16_swap(x, y, false, false);     // {a0, b0, a2, b2}, {a1, b1, a3, b3}
    // The two results go through an opaque asm before and after the second swap: fed straight from one swap builtin into
    // the other, hipcc 7.2 took the second result of v_permlane32_swap to be a copy of the first (tools/kbench/swap_probe.hip
    // shows the instruction itself does what the comment says).
    unsigned r0 = r[0], r1 = r[1];
    asm volatile("" : "+v"(r0), "+v"(r1));
    const auto s = __builtin_amdgcn_permlane32_swap(r0, r1, false, false);   // {a0, b0, a1, b1}, {a2, b2, a3, b3}
    unsigned s0 = s[0], s1 = s[1];
    asm volatile("" : "+v"(s0), "+v"(s1));
    a = __builtin_bit_cast(float, s0);
    b = __builtin_bit_cast(float, s1);
}

template <int NU, int KX2, bool L0, bool GRU, int HEAD = 0>
struct Td16Traits {
    static constexpr int NG = GRU ? 3 : 4;
    static constexpr int KH2 = NU;                          // recurrent k-steps of 32
    static constexpr int BF = 8;                            // bias fragments: 2 row tiles x 4 accumulators
    static constexpr int RF = 2 * NG * KH2;                 // recurrent fragments per unit block
    static constexpr int XS = L0 ? 2 * KX2 - 1 : KX2;       // input k-step groups (first layer: [hi..] per 16 channels, then [lo] of the channels past 16)
    static constexpr int XF = 2 * NG * XS;
    static constexpr int FU = BF + RF + XF;                 // weight fragments per unit block
    static constexpr int NFB = NU * FU;
    static constexpr int HF = HEAD * 2 * NU;                // head fragments: per unit block and channel tile a hi and a lo part
    static constexpr int NF = NFB + HF;
    static constexpr int NS = (NF + TD16_RING_FRAGS - 1) / TD16_RING_FRAGS;
    static constexpr int XFR = 2 * KX2;                     // 1 KiB input fragments per step and batch block
    static constexpr int LDS_W = TD16_RING_DEPTH * TD16_RING_BYTES;
    static constexpr int LDS_X = L0 ? 0 : TD_BLOCK_WAVES * XFR * TD_FRAG_BYTES;
    static constexpr int LDS_TOTAL = LDS_W + LDS_X;
    static constexpr int APD = 8;
    static constexpr int PIECES = TD16_RING_FRAGS / TD_BLOCK_WAVES;
    static_assert(LDS_TOTAL <= 160 * 1024, "LDS budget");
    static_assert(!(HEAD && L0), "the fused head is a variant of the layers above the first");
    static_assert(RF >= 2, "the first input read is issued two slots before the input part");
};

// compile-time instruction schedule of one time step (the scheme of rnn_layer.h: TdSched)
template <int NU, int KX2, bool L0, bool GRU, int HEAD = 0>
struct Td16Sched {
    using T = Td16Traits<NU, KX2, L0, GRU, HEAD>;
    static constexpr int NG = T::NG, FU = T::FU, NF = T::NF, NFB = T::NFB, HF = T::HF, NS = T::NS, APD = T::APD;
    static constexpr int NOPS = GRU ? 7 : 12;   // gate-math micro-ops per accumulator element
    static constexpr int NQ = 16 * NOPS;        // per unit block (16 elements per lane)
    static constexpr int LEAD = TD16_RING_DEPTH - 2;
    static constexpr int PIECE_STEP = TD16_PIECE_STEP;   // a wave's pieces of a stage go out at positions 1, 1 + PIECE_STEP (0: back to back)
    static_assert(T::APD <= TD16_RING_FRAGS, "the LDS read ring must not reach back two stages");

    static constexpr int refill_of(int F) { return F + APD < NF ? F + APD : -1; }
    static constexpr int stage_len(int j) { return j == NS - 1 ? NF - TD16_RING_FRAGS * (NS - 1) : TD16_RING_FRAGS; }
    static constexpr int piece_pos(int j, int i) {
        const int len = stage_len(j);
        return PIECE_STEP * i + 1 < len - 1 ? PIECE_STEP * i + 1 : len - 1;
    }
    static constexpr int wpieces(int F) {
        const int R = refill_of(F);
        if (R < 0) return 0;
        int n = 0;
        for (int i = 0; i < T::PIECES; ++i) n += piece_pos(R / TD16_RING_FRAGS, i) == R % TD16_RING_FRAGS;
        return n;
    }
    // input fragment whose LDS copy is replaced by the next step's in slot F (-1: none).  The last block reads the two
    // tiles of input k-step kx into registers one k-step ahead, so from the k-step's third slot on their LDS copies are dead.
    static constexpr int xdma_of(int F) {
        if (L0 || F >= NFB || F / FU != NU - 1) return -1;
        const int x = F % FU - T::BF - T::RF;
        if (x < 0) return -1;
        const int kx = x / (2 * NG), pos = x % (2 * NG);
        return pos == 2 ? 2 * kx : (pos == 5 ? 2 * kx + 1 : -1);
    }
    static constexpr bool boundary(int F) { return refill_of(F) >= 0 && refill_of(F) % TD16_RING_FRAGS == 0; }
    static constexpr int capacity(int F) {
        if (F / FU == 0) return 0;                       // block 0 of a step has no previous block to finish
        if (F >= NFB && F - NFB >= HF - 2 * HEAD) return 0;   // head slots of the last block follow every one of its micro-ops
        if (boundary(F)) return 0;
        if (wpieces(F) > 0 || xdma_of(F) >= 0) return TD16_CAP_VMEM;
        return TD16_CAP_FREE;
    }
    struct Tab {
        short qbeg[NF], qend[NF];
        short waitn[NS];
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
        if (HF > 0) {
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
        // vmcnt operand of the wait that opens stage j: the vector-memory operations younger than this wave's second piece
        // of that stage (rnn_layer.h, TdSched::make)
        for (int j = 0; j < NS; ++j) {
            int n = 0;
            const int Fb = TD16_RING_FRAGS * j - APD, Fl = TD16_RING_FRAGS * (j - LEAD) + PIECE_STEP + 1 - APD;
#ifndef TD_WAIT_ALL
            if (j >= LEAD && Fl >= 0) {
                for (int F = Fl; F < Fb; ++F) {
                    const int st = (!HEAD && t.qend[F] == NQ && t.qbeg[F] < NQ) ? 2 : 0;   // the block's two output fragments
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

template <int NU, int KX2, bool L0, bool GRU, int HEAD = 0>
__global__ __launch_bounds__(256, 1) void td_rnn_layer16_kernel(TdLayerParams p) {
    using T = Td16Traits<NU, KX2, L0, GRU, HEAD>;
    using S = Td16Sched<NU, KX2, L0, GRU, HEAD>;
    constexpr int NG = T::NG, KH2 = T::KH2, BF = T::BF, RF = T::RF, FU = T::FU, NS = T::NS, NF = T::NF, NFB = T::NFB, HF = T::HF;
    constexpr int APD = T::APD, XFR = T::XFR;
    constexpr int NOPS = S::NOPS, NQ = S::NQ, LEAD = S::LEAD;
    __shared__ __attribute__((aligned(16))) char lds[T::LDS_TOTAL];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int dir = p.dirs == 2 ? (blockIdx.x & 1) : 0;
    const int tile = p.dirs == 2 ? (blockIdx.x >> 1) : blockIdx.x;
    const int bblk = tile * TD_BLOCK_WAVES + wave;
    // a wave past the end of the batch repeats the last block's work: every wave issues the same vector-memory stream,
    // which the counted waits rely on
    const int bb = bblk < p.nbb ? bblk : p.nbb - 1;
#ifdef TD_STAMP
    const uint64_t stamp_c0 = __builtin_amdgcn_s_memtime(), stamp_r0 = __builtin_amdgcn_s_memrealtime();
#endif

    // LDS-DMA pieces from inline asm (see rnn_layer.h: the compiler would drain both counters behind each of them)
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    auto dma = [&](const char *src, unsigned dst) {
#ifdef TD_ABLATE_DMA
        return;
#endif
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(lane16), "s"(src) : "memory");
    };
    const char *wwave = p.wpacked + (size_t)dir * p.dir_stride + (size_t)wave * T::PIECES * TD_FRAG_BYTES;
    const unsigned ldsw = lds0 + wave * T::PIECES * TD_FRAG_BYTES;
    auto issue_piece = [&](const char *wsrc, int stage, int piece, int buf) {
        const char *sp = wsrc + (size_t)stage * TD16_RING_BYTES + piece * TD_FRAG_BYTES;
        asm volatile("" : "+s"(sp));
        dma(sp, ldsw + buf * TD16_RING_BYTES + piece * TD_FRAG_BYTES);
    };
    char *xreg = lds + T::LDS_W + wave * (XFR * TD_FRAG_BYTES);
    const unsigned ldsx = lds0 + T::LDS_W + wave * (XFR * TD_FRAG_BYTES);
    auto xrow = [&](int l) { return p.xin + ((size_t)l * p.nbb + bb) * p.in_frags * TD_FRAG_BYTES; };

    float state[NU][16];   // LSTM: c; GRU: h in fp32.  Element e = 8 t + 4 rt + r of the lane (parked in AGPRs)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) state[u][e] = td16_park(0.f);

    td16_static_for<LEAD>([&](auto jc) {
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int i = 0; i < T::PIECES; ++i) issue_piece(wwave, j % NS, i, j % TD16_RING_DEPTH);
    });
    if constexpr (!L0) {
        const char *x0 = xrow(dir ? p.L - 1 : 0);
#pragma unroll
        for (int i = 0; i < XFR; ++i) dma(x0 + i * TD_FRAG_BYTES, ldsx + i * TD_FRAG_BYTES);
    }
    int ring = TD16_RING_DEPTH - 1;

    // B operand of the bias fragments: ones at k = 0, 1 (lane group 0)
    const int kg = lane >> 4;
    bf16x8 ones = {};
    if (kg == 0) { ones[0] = (__bf16)1.0f; ones[1] = (__bf16)1.0f; }
    bf16x8 ones1 = {};   // first layer: the ones sit in lane group 1, beside x_hi in groups 0 and 2
    if (kg == 1) { ones1[0] = (__bf16)1.0f; ones1[1] = (__bf16)1.0f; }

    f32x4 acc[2][4][2][2];   // [set][accumulator][row tile][sequence tile]
    bf16x8 hold[2][NU];      // h_{t-1}: [sequence tile][unit block = k-step]
    bf16x8 hn[2][NU];        // h_t
    float T0[2][6];
    float hprev = 0.f;
    {
        const bf16x8 zero = {};
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < NU; ++ks) hold[t][ks] = zero;
    }
    const bool out_tiles = TD16_OUT_TILES != 0;   // 1: sequence tiles for a next layer of this kernel, 0: the 32x32x16 fragments

    for (int s = 0; s < p.L; ++s) {
        const int l = dir ? p.L - 1 - s : s;
        const bool last_step = s + 1 == p.L;
        const char *xnext = xrow(last_step ? l : (dir ? l - 1 : l + 1));
        asm volatile("" : "+s"(xnext));
        // first layer: the step's few input fragments stay in registers, swapped into sequence tiles
        bf16x8 xT[L0 ? KX2 : 1][2];
        bf16x8 b2[2];
        if constexpr (L0) {
            const bf16x8 *xp = (const bf16x8 *)xrow(l) + lane;
#pragma unroll
            for (int kx = 0; kx < KX2; ++kx) {
                bf16x8 hi = __builtin_nontemporal_load(xp + (2 * kx) * 64), lo = __builtin_nontemporal_load(xp + (2 * kx + 1) * 64);
                td16_swap16(hi, lo);    // tile 0 / tile 1: [x_hi(0..7) | x_lo(0..7) | x_hi(8..15) | x_lo(8..15)]
                xT[kx][0] = hi;
                xT[kx][1] = lo;
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const bf16x8 zero = {};
                b2[t] = (kg & 1) ? (kg == 1 ? ones1 : zero) : xT[0][t];
            }
        }
        char *op = p.out + (((size_t)l * p.nbb + bb) * p.out_frags + dir * (2 * NU)) * TD_FRAG_BYTES + lane * 16;

        float hs1[2] = {0.f, 0.f}, hs2[2] = {0.f, 0.f};   // HEAD: sum h_t and sum h_t^2 over this lane's units, per sequence tile
        const char *wstep = wwave;
        asm volatile("" : "+s"(wstep));

        auto refill = [&](auto rc) -> bf16x8 {
            constexpr int R = decltype(rc)::value;
            constexpr int stage = R / TD16_RING_FRAGS, pos = R % TD16_RING_FRAGS;
            if constexpr (pos == 0) {
                constexpr int WN = S::tab.waitn[stage];
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(WN) : "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                ring = (ring + 1) & (TD16_RING_DEPTH - 1);
            }
#ifdef TD_ABLATE_LDSREAD
            // diagnostic (tools/kbench): the weight fragments are not read from LDS at all (what the LDS reads cost)
            bf16x8 a = ones;
            asm volatile("" : "+v"(a));
#else
            const bf16x8 a = *(const bf16x8 *)(lds + ring * TD16_RING_BYTES + pos * TD_FRAG_BYTES + lane * 16);
#endif
            td16_static_for<T::PIECES>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (S::piece_pos(stage, i) == pos) {
                    __builtin_amdgcn_sched_barrier(0);
                    issue_piece(wstep, (stage + LEAD) % NS, i, (ring + LEAD) & (TD16_RING_DEPTH - 1));
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            return a;
        };
        auto xfrag = [&](int i) -> bf16x8 { return *(const bf16x8 *)(xreg + i * TD_FRAG_BYTES + lane * 16); };

        // micro-op q of block ub's gate nonlinearities (the instruction sequence of rnn_layer.h's gate_op; element
        // e = 2 pair + (w & 1) = 8 t + 4 rt + r, elements go in pairs that are neighbours in the bf16 operand)
        auto gate_op = [&](auto ubc, auto qc) {
            constexpr int ub = decltype(ubc)::value, q = decltype(qc)::value;
            constexpr int pair = q / (2 * NOPS), w = q % (2 * NOPS), o = w / 2, e = 2 * pair + (w & 1);
            constexpr int et = e >> 3, ert = (e >> 2) & 1, er = e & 3;
            constexpr float P2L2E = TD_GATE_SCALE_TANH;
            f32x4(&a)[4][2][2] = acc[ub & 1];
            float(&t)[6] = T0[e & 1];
            float hval = 0.f;
#ifdef TD_ABLATE_GATES
            if constexpr (o == NOPS - 1) hval = a[0][ert][et][er];
            if constexpr (false) {
#else
            if constexpr (!GRU) {
#endif
                if constexpr (o == 0) t[0] = __builtin_amdgcn_exp2f(a[0][ert][et][er]);
                if constexpr (o == 1) t[1] = __builtin_amdgcn_exp2f(a[1][ert][et][er]);
                if constexpr (o == 2) t[2] = __builtin_amdgcn_exp2f(a[2][ert][et][er]);
                if constexpr (o == 3) t[3] = __builtin_amdgcn_exp2f(a[3][ert][et][er]);
                if constexpr (o == 4) t[0] = __builtin_amdgcn_rcpf(1.0f + t[0]);  // i
                if constexpr (o == 5) t[1] = __builtin_amdgcn_rcpf(1.0f + t[1]);  // f
                if constexpr (o == 6) t[2] = __builtin_amdgcn_rcpf(t[2] + 1.0f);  // g = 1 - 2 t[2]
                if constexpr (o == 7) t[3] = __builtin_amdgcn_rcpf(1.0f + t[3]);  // o
                if constexpr (o == 8) {
                    const float gg = __builtin_fmaf(-2.0f, t[2], 1.0f);
                    const float cv = __builtin_fmaf(t[1], td16_unpark(state[ub][e]), t[0] * gg);
                    state[ub][e] = td16_park(cv);
                    t[4] = cv;
                }
                if constexpr (o == 9) t[5] = __builtin_amdgcn_exp2f(P2L2E * t[4]);
                if constexpr (o == 10) t[5] = __builtin_amdgcn_rcpf(t[5] + 1.0f);
                if constexpr (o == 11) hval = t[3] * __builtin_fmaf(-2.0f, t[5], 1.0f);
            } else {
                if constexpr (o == 0) t[0] = __builtin_amdgcn_exp2f(a[0][ert][et][er]);
                if constexpr (o == 1) t[1] = __builtin_amdgcn_exp2f(a[1][ert][et][er]);
                if constexpr (o == 2) t[0] = __builtin_amdgcn_rcpf(1.0f + t[0]);  // r
                if constexpr (o == 3) t[1] = __builtin_amdgcn_rcpf(1.0f + t[1]);  // z
                if constexpr (o == 4) t[2] = __builtin_amdgcn_exp2f(__builtin_fmaf(t[0], a[3][ert][et][er], a[2][ert][et][er]));
                if constexpr (o == 5) t[2] = __builtin_amdgcn_rcpf(t[2] + 1.0f);
                if constexpr (o == 6) {
                    const float ng = __builtin_fmaf(-2.0f, t[2], 1.0f);
                    const float h = __builtin_fmaf(t[1], td16_unpark(state[ub][e]), (1.0f - t[1]) * ng);
                    state[ub][e] = td16_park(h);
                    hval = h;
                }
            }
            if constexpr (o == NOPS - 1) {
                if constexpr ((e & 1) == 0) hprev = hval;
                else {
                    const __bf16 b0 = (__bf16)hprev, b1 = (__bf16)hval;
                    hn[et][ub][4 * ert + er - 1] = b0;
                    hn[et][ub][4 * ert + er] = b1;
                    if constexpr (HEAD > 0) {
                        const float f0 = (float)b0, f1 = (float)b1;
                        hs1[et] += f0 + f1;
                        hs2[et] = __builtin_fmaf(f0, f0, __builtin_fmaf(f1, f1, hs2[et]));
                    }
                }
            }
            if constexpr (q == NQ - 1 && HEAD == 0) {
                // h_t of this block goes to HBM right away (streaming stores): as the two sequence tiles for a next layer
                // of this family, swapped into the F pair (k-steps 2 ub, 2 ub + 1) for every other reader
                bf16x8 o0 = hn[0][ub], o1 = hn[1][ub];
                if (!out_tiles) td16_swap16(o0, o1);
                __builtin_nontemporal_store(o0, (bf16x8 *)(op + (2 * ub) * TD_FRAG_BYTES));
                __builtin_nontemporal_store(o1, (bf16x8 *)(op + (2 * ub + 1) * TD_FRAG_BYTES));
            }
        };

        bf16x8 A[APD];
        td16_static_for<(APD < NF ? APD : NF)>([&](auto ic) { A[decltype(ic)::value] = refill(ic); });
        bf16x8 xb[2][2];   // input operands [k-step parity][sequence tile], read from LDS one k-step ahead
        const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

        td16_static_for<NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            td16_static_for<FU>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                constexpr int F = u * FU + f;
                if constexpr (!L0 && f == BF + RF - 2) { xb[0][0] = xfrag(0); xb[0][1] = xfrag(1); }
                const bf16x8 a = A[F % APD];
                f32x4(&ac)[4][2][2] = acc[u & 1];
                if constexpr (f < BF) {
                    // bias (LSTM: b_ih + b_hh per gate; GRU: r, z, n-input, n-hidden): the first touch of the accumulators
                    constexpr int rt = f / 4, ai = f % 4;
                    if constexpr (L0) {
                        ac[ai][rt][0] = TD16_MFMA(a, b2[0], zero4);
                        ac[ai][rt][1] = TD16_MFMA(a, b2[1], zero4);
                    } else {
                        ac[ai][rt][0] = TD16_MFMA(a, ones, zero4);
                        ac[ai][rt][1] = TD16_MFMA(a, ones, zero4);
                    }
                } else if constexpr (f - BF < RF) {
                    constexpr int idx = f - BF, ks = idx / (2 * NG), rt = (idx / NG) % 2, g = idx % NG;
                    constexpr int ai = (GRU && g == 2) ? 3 : g;
                    ac[ai][rt][0] = TD16_MFMA(a, hold[0][ks], ac[ai][rt][0]);
                    ac[ai][rt][1] = TD16_MFMA(a, hold[1][ks], ac[ai][rt][1]);
                } else if constexpr (L0) {
                    constexpr int x = f - BF - RF, sec = x / (2 * NG), rt = (x / NG) % 2, g = x % NG;
                    constexpr int kx = sec < KX2 ? sec : sec - KX2 + 1;   // [W_hi ...] per 16 channels, then [W_lo | 0 ...] of the channels past 16
                    ac[g][rt][0] = TD16_MFMA(a, xT[kx][0], ac[g][rt][0]);
                    ac[g][rt][1] = TD16_MFMA(a, xT[kx][1], ac[g][rt][1]);
                } else {
                    constexpr int x = f - BF - RF, kx = x / (2 * NG), rt = (x / NG) % 2, g = x % NG;
                    if constexpr (rt == 0 && g == 0 && kx + 1 < KX2) {
                        xb[(kx + 1) % 2][0] = xfrag(2 * (kx + 1));
                        xb[(kx + 1) % 2][1] = xfrag(2 * (kx + 1) + 1);
                    }
                    ac[g][rt][0] = TD16_MFMA(a, xb[kx % 2][0], ac[g][rt][0]);
                    ac[g][rt][1] = TD16_MFMA(a, xb[kx % 2][1], ac[g][rt][1]);
                }
#ifdef TD16_PAIR_PIN
                __builtin_amdgcn_sched_barrier(0);   // the slot's two matrix instructions stay adjacent
#endif
                if constexpr (S::refill_of(F) >= 0)
                    A[F % APD] = refill(std::integral_constant<int, (S::refill_of(F) >= 0 ? S::refill_of(F) : 0)>{});
                if constexpr (S::xdma_of(F) >= 0) {
                    constexpr int xi = S::xdma_of(F) >= 0 ? S::xdma_of(F) : 0;
                    __builtin_amdgcn_sched_barrier(0);
                    dma(xnext + xi * TD_FRAG_BYTES, ldsx + xi * TD_FRAG_BYTES);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (u > 0) {
                    constexpr int q0 = S::tab.qbeg[F], q1 = S::tab.qend[F];
                    td16_static_for<q1 - q0>([&](auto jc) {
                        gate_op(std::integral_constant<int, (u > 0 ? u - 1 : 0)>{}, std::integral_constant<int, q0 + decltype(jc)::value>{});
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        });

        if constexpr (HEAD == 0) {
            // the last block's gates have no MFMAs left to hide behind
            td16_static_for<NQ>([&](auto qc) { gate_op(std::integral_constant<int, NU - 1>{}, qc); });
        } else {
            // head slots: partial projection of h_t, unit block by unit block (per channel tile the hi and the lo part of the
            // folded weights); the blocks before the last run beside the last block's gate math, its own slots after it
            f32x4(&hacc)[2][2] = acc[NU & 1][0];   // [channel tile][sequence tile], in the accumulator set the last block does not use
            if constexpr (NU == 1) td16_static_for<NQ>([&](auto qc) { gate_op(std::integral_constant<int, NU - 1>{}, qc); });
            td16_static_for<HF>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                constexpr int F = NFB + f;
                constexpr int ub = f / (2 * HEAD), ct = (f / 2) % HEAD;
                const bf16x8 a = A[F % APD];
                if constexpr (ub == 0 && f % 2 == 0) {
                    hacc[ct][0] = TD16_MFMA(a, hn[0][ub], zero4);
                    hacc[ct][1] = TD16_MFMA(a, hn[1][ub], zero4);
                } else {
                    hacc[ct][0] = TD16_MFMA(a, hn[0][ub], hacc[ct][0]);
                    hacc[ct][1] = TD16_MFMA(a, hn[1][ub], hacc[ct][1]);
                }
                if constexpr (S::refill_of(F) >= 0)
                    A[F % APD] = refill(std::integral_constant<int, (S::refill_of(F) >= 0 ? S::refill_of(F) : 0)>{});
                if constexpr (NU > 1) {
                    constexpr int q0 = S::tab.qbeg[F], q1 = S::tab.qend[F];
                    td16_static_for<q1 - q0>([&](auto jc) {
                        gate_op(std::integral_constant<int, NU - 1>{}, std::integral_constant<int, q0 + decltype(jc)::value>{});
                    });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
            // into the arrangement td_head_combine_kernel reads (the 32x32 accumulator's): row r of lane (m, hf) = output
            // channel (r & 3) + 8 (r >> 2) + 4 hf of sequence m; then sum h, sum h^2 (lanes m, m + 32 hold two halves)
            float *hp = p.hpart + ((((size_t)l * p.nbb + bb) * p.dirs + dir) * 64 + lane) * p.hp_stride;
#pragma unroll
            for (int ct = 0; ct < HEAD; ++ct) {
                f32x4 lo = hacc[ct][0], hi = hacc[ct][1];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float x = lo[r], y = hi[r];
                    td16_to_rows32(x, y);
                    lo[r] = x;
                    hi[r] = y;
                }
                *(f32x4 *)(hp + 8 * ct) = lo;
                *(f32x4 *)(hp + 8 * ct + 4) = hi;
            }
            td16_to_rows32(hs1[0], hs1[1]);
            td16_to_rows32(hs2[0], hs2[1]);
            *(f32x4 *)(hp + p.hp_stride - 4) = f32x4{hs1[0] + hs1[1], hs2[0] + hs2[1], 0.f, 0.f};
        }

#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int ks = 0; ks < NU; ++ks) hold[t][ks] = hn[t][ks];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef TD_STAMP
    if (tid == 0) {
        unsigned long long *st = (unsigned long long *)p.save + 2 * blockIdx.x;
        st[0] = __builtin_amdgcn_s_memtime() - stamp_c0;
        st[1] = __builtin_amdgcn_s_memrealtime() - stamp_r0;
    }
#endif
}

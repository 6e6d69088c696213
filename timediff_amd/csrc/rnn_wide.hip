// The unit-split recurrent layer: wide hidden sizes (256 < H <= 1024: BASELINE.json configs[3], [4]) and small
// batches at any width of two or more 32-unit blocks (rnn.hip: td_use_wide), plus the head for wide layers.
//
// Above H = 256 the state of 32 sequences (cell state fp32 + h as B fragments) no longer fits one wave's
// registers, so the decomposition of rnn_layer.h turns by 90 degrees: a workgroup owns NB batch blocks of 32
// sequences for all L steps and its WAVES waves split the hidden units: wave w owns the NUW 32-unit blocks
// w*NUW .. w*NUW+NUW-1, for every batch block, so a weight fragment fetched from L2 feeds NB MFMAs.
//   NB = 2, WAVES = 4   the batch fills the chip with two blocks per workgroup (more than 256 single-block workgroups)
//   NB = 1, WAVES = 4   fewer workgroups than CUs: one block each
//   NB = 1, WAVES = 8   at most 32 workgroups (the reference's batch sizes): a second wave per SIMD hides L2 latency
//   NB = 1, WAVES = 2, 3, 6 (one block per wave)   widths of 2, 3 or 6 unit blocks (HiRID's 36, MIMIC-III's 92) at small batches
// The fp32 recurrent state of a block lives in L2/HBM between steps (read under the block's last MFMAs, 8 H bytes
// per sequence and step: nothing next to the weight stream), which leaves the registers to the accumulators and
// the operand pipeline.  h_{t-1} of all units is the B operand of every wave's recurrent k-steps and lives in LDS
// (NB * KH KiB; double buffered when two copies fit, else h_t goes to the layer output first and is copied back
// between two barriers).  The weight fragments are the same packed stream rnn_layer.h reads ([unit block][bias |
// W_hh k-steps | W_ih k-steps][gate], td_pack_weights_kernel); each wave streams its own blocks' fragments
// L2 -> registers through a pinned two-chunk software pipeline (a chunk = KC k-steps = KC*NG fragments).  The
// layer's input fragments are read from L2/HBM as B operands with the chunk that uses them.
//
// Gate math: the same instruction sequence as rnn_layer.h's gate_op (the accumulators arrive pre-multiplied
// by the exp2 factors), so the oracle's bf16 emulation covers both kernels.
#include "rnn_geom.h"
#include "rnn_params.h"

template <int N, class F>
__device__ __forceinline__ void tdw_static_for(F &&f) {
    if constexpr (N > 0) {
        tdw_static_for<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

#ifndef TDW_EIGHT_WAVES_MAX_WG
#define TDW_EIGHT_WAVES_MAX_WG 32   // up to this many workgroups (512 sequences bidirectional) the small-batch variant runs eight waves each
#endif

// NB: batch blocks (of 32 sequences) per workgroup, every weight fragment feeds NB MFMAs.  2 when the batch fills
// the chip that way, 1 for small batches (twice the workgroups).
// WAVES: waves per workgroup = ways the unit blocks are split: 4 (one wave per SIMD), or 8 for small batches, where
// only a few workgroups run and a second wave per SIMD hides the other's L2 latency.
template <int NUW, int NB, bool L0, bool GRU, bool SAVE, int WAVES = 4>
__global__ __launch_bounds__(64 * WAVES, 1) void td_rnn_layer_wide_kernel(TdLayerParams p, int kx0) {
    constexpr int NG = GRU ? 3 : 4, NU = WAVES * NUW, KH = 2 * NU;
    // k-steps / weight fragments per pipeline chunk: 4 k-steps where a wave has 512 registers and KH allows an even
    // number of chunks, fewer for the 6- and 8-wave variants (256 registers per wave) and the narrow widths
    constexpr int KC = (WAVES >= 6) ? 2 : (KH % 8 == 0 ? 4 : (KH % 4 == 0 ? 2 : 1)), CH = KC * NG;
    static_assert((KH / KC) % 2 == 0, "the two-chunk pipeline needs an even number of chunks per part");
    // h_{t-1} as B fragments, [NB][KH] KiB.  Double buffered when two copies fit (one barrier per step, h_t written
    // straight into the other buffer); otherwise single buffered: h_t goes to the layer output first and is copied
    // back into LDS between two barriers.
    constexpr bool DBUF = NB * NUW * WAVES <= 32;
    extern __shared__ __attribute__((aligned(16))) char hlds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int dir = p.dirs == 2 ? (blockIdx.x & 1) : 0;
    const int tile = p.dirs == 2 ? (blockIdx.x >> 1) : blockIdx.x;
    // an odd batch-block count leaves the last tile one block short: it repeats its first block (same inputs, same
    // stores), so every wave runs the same instruction stream
    int bbs[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) bbs[n] = tile * NB + n < p.nbb ? tile * NB + n : p.nbb - 1;
    const int KXF = L0 ? 0 : p.in_frags;                 // input k-steps above the first layer (= dirs * KH)
    const int XFR0 = L0 ? 2 * kx0 : 0;                   // first layer: hi + lo fragments
    const int FU = 4 + (KH + (L0 ? XFR0 : KXF)) * NG;    // weight fragments per unit block

    for (int i = tid; i < NB * KH * 64; i += 64 * WAVES) ((bf16x8 *)hlds)[i] = bf16x8{};   // h_{-1} = 0 (buffer 0)
    __syncthreads();

    bf16x8 ones = {};
    if (lane < 32) { ones[0] = (__bf16)1.0f; ones[1] = (__bf16)1.0f; }
    const bf16x8 *wdir = (const bf16x8 *)(p.wpacked + (size_t)dir * p.dir_stride) + lane;

    for (int s = 0; s < p.L; ++s) {
        const int l = dir ? p.L - 1 - s : s;
        const bf16x8 *hl = (const bf16x8 *)(hlds + (DBUF ? (size_t)(s & 1) * NB * KH * TD_FRAG_BYTES : 0)) + lane;
        bf16x8 *hnext = (bf16x8 *)(hlds + (size_t)((s + 1) & 1) * NB * KH * TD_FRAG_BYTES) + lane;   // DBUF only
        const bf16x8 *xp[NB];
        bf16x8 *op[NB];
        float *stp[NB];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            xp[n] = (const bf16x8 *)(p.xin + ((size_t)l * p.nbb + bbs[n]) * p.in_frags * TD_FRAG_BYTES) + lane;
            op[n] = (bf16x8 *)(p.out + (((size_t)l * p.nbb + bbs[n]) * p.out_frags + dir * KH) * TD_FRAG_BYTES) + lane;
            stp[n] = p.state + (size_t)dir * p.state_dir_stride + (size_t)bbs[n] * NU * 1024 + lane * 16;
        }

#pragma unroll 1
        for (int i = 0; i < NUW; ++i) {
            const int u = wave * NUW + i;
            const bf16x8 *wa = wdir + (size_t)u * FU * 64;
            // recurrent state of this block (LSTM: c, GRU: h, fp32): it lives in L2 / HBM between steps -- 8 H bytes
            // per sequence and step, nothing next to the weight stream -- and is read here, a whole block of MFMAs
            // ahead of its use (under the block's last chunks), so the register file keeps room for a second batch block's accumulators
            f32x4 cst[NB][4];
            auto ldState = [&]() {   // issued under the block's last two chunks of MFMAs
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        cst[n][q] = s > 0 ? *(const f32x4 *)(stp[n] + (size_t)u * 1024 + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
            };
            f32x16 acc[NB][4];
#pragma unroll
            for (int n = 0; n < NB; ++n)
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[n][g][r] = 0.f;

            // Two-chunk software pipeline over the weight stream after the bias fragments; a chunk = KC k-steps =
            // CH weight fragments + the B operands they meet (h_{t-1} from LDS, then the layer input from L2).
            // Straight-line bodies (no branch between a load and its use), so the compiler's counted vmcnt waits
            // let chunk c+1's loads stay in flight under chunk c's MFMAs.
            bf16x8 A0[CH], A1[CH], B0[NB][KC], B1[NB][KC];
            auto ldA = [&](int c, bf16x8(&A)[CH]) {
                const bf16x8 *wp = wa + (size_t)(4 + c * CH) * 64;
#pragma unroll
                for (int j = 0; j < CH; ++j) A[j] = wp[j * 64];
            };
            auto ldH = [&](int c, bf16x8(&B)[NB][KC]) {
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int k = 0; k < KC; ++k) B[n][k] = hl[(n * KH + KC * c + k) * 64];
            };
            auto ldX = [&](int c, bf16x8(&B)[NB][KC]) {   // c counts from the first input chunk
#pragma unroll
                for (int n = 0; n < NB; ++n)
#pragma unroll
                    for (int k = 0; k < KC; ++k) B[n][k] = xp[n][(KC * c + k) * 64];
            };
            auto mma = [&](auto recc, bf16x8(&A)[CH], bf16x8(&B)[NB][KC]) {
                constexpr bool rec = decltype(recc)::value;
#pragma unroll
                for (int k = 0; k < KC; ++k)
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        // GRU n gate: the recurrent half accumulates apart (it is multiplied by r)
                        const int ai = (GRU && g == 2 && rec) ? 3 : g;
#pragma unroll
                        for (int n = 0; n < NB; ++n)
                            acc[n][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k * NG + g], B[n][k], acc[n][ai], 0, 0, 0);
                    }
            };
            constexpr std::true_type REC{};
            constexpr std::false_type INP{};
            const int NH = KH / KC, NX = KXF / KC;   // both even
            // PIN: the machine scheduler otherwise sinks each load next to its MFMA to save registers, which turns the
            // pipeline into load -> wait -> 2 MFMAs per fragment
#define TDW_PIN() __builtin_amdgcn_sched_barrier(0)
            ldA(0, A0); ldH(0, B0);
            TDW_PIN();
            // bias k-step (LSTM: b_ih + b_hh per gate; GRU: r, z, n-input, n-hidden), hi + lo columns against ones
#pragma unroll
            for (int f = 0; f < 4; ++f) {
                const bf16x8 wb = wa[f * 64];
#pragma unroll
                for (int n = 0; n < NB; ++n) acc[n][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, ones, acc[n][f], 0, 0, 0);
            }
#pragma unroll 1
            for (int c = 0; c + 2 < NH; c += 2) {
                TDW_PIN(); ldA(c + 1, A1); ldH(c + 1, B1); TDW_PIN();
                mma(REC, A0, B0);
                TDW_PIN(); ldA(c + 2, A0); ldH(c + 2, B0); TDW_PIN();
                mma(REC, A1, B1);
            }
            TDW_PIN(); ldA(NH - 1, A1); ldH(NH - 1, B1); TDW_PIN();
            mma(REC, A0, B0);
            if constexpr (!L0) {
                TDW_PIN(); ldA(NH, A0); ldX(0, B0); TDW_PIN();   // hand over to the input part without draining the pipeline
                mma(REC, A1, B1);
#pragma unroll 1
                for (int c = 0; c + 2 < NX; c += 2) {
                    TDW_PIN(); ldA(NH + c + 1, A1); ldX(c + 1, B1); TDW_PIN();
                    mma(INP, A0, B0);
                    TDW_PIN(); ldA(NH + c + 2, A0); ldX(c + 2, B0); TDW_PIN();
                    mma(INP, A1, B1);
                }
                TDW_PIN(); ldA(NH + NX - 1, A1); ldX(NX - 1, B1); TDW_PIN();
                mma(INP, A0, B0);
                TDW_PIN(); ldState(); TDW_PIN();
                mma(INP, A1, B1);
            } else {
                TDW_PIN(); ldState(); TDW_PIN();
                mma(REC, A1, B1);
                // first layer: W = Whi + Wlo, x = xhi + xlo in bf16, the three leading products (rnn_layer.h)
                const bf16x8 *wp = wa + (size_t)(4 + KH * NG) * 64;
                for (int kx = 0; kx < kx0; ++kx) {
                    bf16x8 xhi[NB], xlo[NB];
#pragma unroll
                    for (int n = 0; n < NB; ++n) { xhi[n] = xp[n][(2 * kx) * 64]; xlo[n] = xp[n][(2 * kx + 1) * 64]; }
#pragma unroll
                    for (int g = 0; g < NG; ++g) {
                        const bf16x8 whi = wp[(size_t)(2 * (kx * NG + g)) * 64], wlo = wp[(size_t)(2 * (kx * NG + g) + 1) * 64];
#pragma unroll
                        for (int n = 0; n < NB; ++n) {
                            acc[n][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, xlo[n], acc[n][g], 0, 0, 0);
                            acc[n][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, xhi[n], acc[n][g], 0, 0, 0);
                            acc[n][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xhi[n], acc[n][g], 0, 0, 0);
                        }
                    }
                }
            }

            // gates (row r of the accumulator = unit 32u + (r&3) + 8(r>>2) + 4hf)
            constexpr float P2L2E = TD_GATE_SCALE_TANH;
#pragma unroll
            for (int n = 0; n < NB; ++n) {
                float hv[16];
                float sv[5][16];
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float prev = cst[n][r >> 2][r & 3];
                    if constexpr (!GRU) {
                        const float e0 = __builtin_amdgcn_exp2f(acc[n][0][r]), e1 = __builtin_amdgcn_exp2f(acc[n][1][r]);
                        const float e2 = __builtin_amdgcn_exp2f(acc[n][2][r]), e3 = __builtin_amdgcn_exp2f(acc[n][3][r]);
                        const float ig = __builtin_amdgcn_rcpf(1.0f + e0), fg = __builtin_amdgcn_rcpf(1.0f + e1);
                        const float t2 = __builtin_amdgcn_rcpf(e2 + 1.0f), og = __builtin_amdgcn_rcpf(1.0f + e3);
                        const float gg = __builtin_fmaf(-2.0f, t2, 1.0f);
                        const float cv = __builtin_fmaf(fg, prev, ig * gg);
                        cst[n][r >> 2][r & 3] = cv;
                        float t5 = __builtin_amdgcn_exp2f(P2L2E * cv);
                        t5 = __builtin_amdgcn_rcpf(t5 + 1.0f);
                        hv[r] = og * __builtin_fmaf(-2.0f, t5, 1.0f);
                        if constexpr (SAVE) { sv[0][r] = ig; sv[1][r] = fg; sv[2][r] = gg; sv[3][r] = og; sv[4][r] = cv; }
                    } else {
                        const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[n][0][r]));
                        const float zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[n][1][r]));
                        float t2 = __builtin_amdgcn_exp2f(__builtin_fmaf(rg, acc[n][3][r], acc[n][2][r]));
                        t2 = __builtin_amdgcn_rcpf(t2 + 1.0f);
                        const float ng = __builtin_fmaf(-2.0f, t2, 1.0f);
                        const float h = __builtin_fmaf(zg, prev, (1.0f - zg) * ng);
                        cst[n][r >> 2][r & 3] = h;
                        hv[r] = h;
                        if constexpr (SAVE) { sv[0][r] = rg; sv[1][r] = zg; sv[2][r] = ng; sv[3][r] = acc[n][3][r] * (1.0f / P2L2E); sv[4][r] = h; }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) *(f32x4 *)(stp[n] + (size_t)u * 1024 + 4 * q) = cst[n][q];
                bf16x8 h0, h1;
#pragma unroll
                for (int j = 0; j < 8; ++j) { h0[j] = (__bf16)hv[j]; h1[j] = (__bf16)hv[8 + j]; }
                // h_t goes to the layer output now and from there into LDS once every wave is done with h_{t-1}
                if constexpr (DBUF) {
                    // the other buffer was last read in step s - 1, which every wave has left
                    hnext[(n * KH + 2 * u) * 64] = h0;
                    hnext[(n * KH + 2 * u + 1) * 64] = h1;
                    __builtin_nontemporal_store(h0, op[n] + (2 * u) * 64);
                    __builtin_nontemporal_store(h1, op[n] + (2 * u + 1) * 64);
                } else {
                    op[n][(2 * u) * 64] = h0;
                    op[n][(2 * u + 1) * 64] = h1;
                }
                if constexpr (SAVE) {
                    float *svb = p.save + (size_t)dir * p.save_dir_stride + ((((size_t)l * p.nbb + bbs[n]) * NU + u) * 5) * 1024 + lane * 16;
#pragma unroll
                    for (int k = 0; k < 5; ++k)
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            *(f32x4 *)(svb + k * 1024 + 4 * q) = f32x4{sv[k][4 * q], sv[k][4 * q + 1], sv[k][4 * q + 2], sv[k][4 * q + 3]};
                }
            }
        }
        if constexpr (!DBUF) {
            __syncthreads();   // every wave is done with h_{t-1} (and this wave's stores of h_t have been issued and waited for)
            // this wave's part of h_t: back from the layer output (its own stores, L2) into the LDS buffer
#pragma unroll
            for (int n = 0; n < NB; ++n)
                for (int f = 0; f < 2 * NUW; ++f) {
                    const int fr = 2 * wave * NUW + f;
                    ((bf16x8 *)hlds)[(size_t)(n * KH + fr) * 64 + lane] = op[n][fr * 64];
                }
        }
        __syncthreads();   // h_t complete in LDS
    }
}

template <int NUW, int NB, bool L0, bool GRU, bool SAVE, int WAVES = 4>
static int tdw_launch(const TdLayerParams &p, int kx0, hipStream_t stream) {
    const size_t shm = (size_t)(NB * NUW * WAVES <= 32 ? 2 : 1) * NB * (2 * WAVES * NUW) * TD_FRAG_BYTES;
    static bool attr_done = false;   // per instantiation
    if (!attr_done) {
        TD_HIP(hipFuncSetAttribute((const void *)td_rnn_layer_wide_kernel<NUW, NB, L0, GRU, SAVE, WAVES>,
                                   hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        attr_done = true;
    }
    const int tiles = td_cdiv(p.nbb, NB);
    hipLaunchKernelGGL((td_rnn_layer_wide_kernel<NUW, NB, L0, GRU, SAVE, WAVES>), dim3(tiles * p.dirs), dim3(64 * WAVES), shm,
                       stream, p, kx0);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

template <int NUW, int NB, int WAVES>
static int tdw_dispatch2(int gru, int save, const TdLayerParams &p, bool layer0, int kx0, hipStream_t stream) {
    if (layer0) {
        if (!gru) return save ? tdw_launch<NUW, NB, true, false, true, WAVES>(p, kx0, stream) : tdw_launch<NUW, NB, true, false, false, WAVES>(p, kx0, stream);
        return save ? tdw_launch<NUW, NB, true, true, true, WAVES>(p, kx0, stream) : tdw_launch<NUW, NB, true, true, false, WAVES>(p, kx0, stream);
    }
    if (!gru) return save ? tdw_launch<NUW, NB, false, false, true, WAVES>(p, kx0, stream) : tdw_launch<NUW, NB, false, false, false, WAVES>(p, kx0, stream);
    return save ? tdw_launch<NUW, NB, false, true, true, WAVES>(p, kx0, stream) : tdw_launch<NUW, NB, false, true, false, WAVES>(p, kx0, stream);
}

// ---------------------------------------------------------------------------------------------
// The same layer with ONE launch per time step (td_use_step_launch, rnn_geom.h): wide layers at small batches.
// The persistent kernel above gives a batch block to one workgroup, which pulls the direction's whole weight stream
// (6.2 MB at H = 512, 25 MB at H = 1024) through ONE CU's L2 port every step: 77 us per step at H = 512, whatever the
// batch, with 32 of 256 CUs busy at batch 512.  Here a wave owns one (unit block, batch block[s], direction) of one
// step: nu * dirs * batch blocks waves per launch spread over all CUs, each streaming only its unit block's slice
// (388 KiB at H = 512); the waves of a workgroup work the SAME unit block for different batch blocks, and a slice's
// workgroups sit on one XCD (see the index arithmetic at the top of the kernel).  h_{t-1}
// is read from the layer's own output of the step before (a kernel boundary lies between its stores and these loads:
// no hand-off protocol), the fp32 state from the same scratch as above.  A step costs a kernel boundary (4-5 us)
// instead of a barrier, which is why the persistent form stays for batches that fill the chip.
// Same bias / recurrent / input order of the MFMAs and the same gate sequence as the kernel above: bit-identical results.
// ---------------------------------------------------------------------------------------------
template <int NB, bool L0, bool GRU, bool SAVE>
__global__ __launch_bounds__(256, 1) void td_rnn_step_wide_kernel(TdLayerParams p, int kx0, int s, int nu) {
    constexpr int NG = GRU ? 3 : 4, KC = 4, CH = KC * NG;
    constexpr int D = NB == 1 ? 4 : 2;          // chunks in the ring (80 registers each at NB = 1)
    const int KH = 2 * nu;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6;
    // Workgroups go to the XCDs round-robin by index, and after a kernel boundary an XCD's L2 is cold: what a step reads comes
    // from the Infinity Cache once per XCD that touches it.  An XCD therefore works ONE direction and every G-th unit block
    // of it (G = 8 / dirs XCDs per direction) for all batch blocks: nu / G weight slices + that direction's h_{t-1} and x_t.
    // (Measured at H = 512, batch 512: no difference against the plain (unit block, direction) order; kept for H = 1024.)
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int G = 8 / p.dirs, ntile = (p.nbb + NB - 1) / NB, ngrp = (ntile + nw - 1) / nw;
    const int dir = p.dirs == 2 ? xcd / 4 : 0;
    const int u = xcd % G + G * (slot / ngrp);
    const int tile = (slot % ngrp) * nw + wave;
    if (u >= nu || tile >= ntile) return;                       // (no barrier in this kernel)
    int bbs[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) bbs[n] = tile * NB + n < p.nbb ? tile * NB + n : p.nbb - 1;   // an odd count: the last block twice (same values)
    const int KXF = L0 ? 0 : p.in_frags;
    const int XFR0 = L0 ? 2 * kx0 : 0;
    const int FU = 4 + (KH + (L0 ? XFR0 : KXF)) * NG;
    const int l = dir ? p.L - 1 - s : s, lp = dir ? l + 1 : l - 1;

    bf16x8 ones = {};
    if (lane < 32) { ones[0] = (__bf16)1.0f; ones[1] = (__bf16)1.0f; }
    const bf16x8 *wa = (const bf16x8 *)(p.wpacked + (size_t)dir * p.dir_stride) + lane + (size_t)u * FU * 64;
    const bf16x8 *xp[NB], *hl[NB];
    bf16x8 *op[NB];
    float *stp[NB];
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        xp[n] = (const bf16x8 *)(p.xin + ((size_t)l * p.nbb + bbs[n]) * p.in_frags * TD_FRAG_BYTES) + lane;
        hl[n] = (const bf16x8 *)(p.out + (((size_t)(s > 0 ? lp : l) * p.nbb + bbs[n]) * p.out_frags + dir * KH) * TD_FRAG_BYTES) + lane;
        op[n] = (bf16x8 *)(p.out + (((size_t)l * p.nbb + bbs[n]) * p.out_frags + dir * KH) * TD_FRAG_BYTES) + lane;
        stp[n] = p.state + (size_t)dir * p.state_dir_stride + ((size_t)bbs[n] * nu + u) * 1024 + lane * 16;
    }
    f32x4 cst[NB][4];
    auto ldState = [&]() {
#pragma unroll
        for (int n = 0; n < NB; ++n)
#pragma unroll
            for (int q = 0; q < 4; ++q) cst[n][q] = s > 0 ? *(const f32x4 *)(stp[n] + 4 * q) : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    f32x16 acc[NB][4];
#pragma unroll
    for (int n = 0; n < NB; ++n)
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[n][g][r] = 0.f;
    // A ring of D chunks (a chunk = KC k-steps = CH weight fragments + the B operands they meet), D - 1 of them in flight
    // while one is multiplied: with a few waves per CU a chunk's 20 KiB take ~1 us to arrive from L2 and its 16 MFMAs
    // 0.25 us.  The recurrent and the input k-steps are one stream of chunks (the weight fragments are contiguous, the B
    // operand's source is selected by pointer); at s = 0 the recurrent chunks are skipped (h_{-1} = 0 adds exact zeros).
    // Straight-line bodies: the loads past the last chunk re-read it instead of branching.
    bf16x8 A[D][CH], B[D][NB][KC];
    const int NH = KH / KC, NX = KXF / KC, NT = NH + NX;
    auto issue = [&](int c, auto jc) {
        constexpr int j = decltype(jc)::value;
        const int cc = c < NT ? c : NT - 1;
        const bf16x8 *wp = wa + (size_t)(4 + cc * CH) * 64;
#pragma unroll
        for (int i = 0; i < CH; ++i) A[j][i] = wp[i * 64];
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const bf16x8 *bp = cc < NH ? hl[n] + (size_t)(KC * cc) * 64 : xp[n] + (size_t)(KC * (cc - NH)) * 64;
#pragma unroll
            for (int k = 0; k < KC; ++k) B[j][n][k] = bp[k * 64];
        }
    };
    auto mma = [&](auto recc, auto jc) {
        constexpr bool rec = decltype(recc)::value;
        constexpr int j = decltype(jc)::value;
#pragma unroll
        for (int k = 0; k < KC; ++k)
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int ai = (GRU && g == 2 && rec) ? 3 : g;
#pragma unroll
                for (int n = 0; n < NB; ++n)
                    acc[n][ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[j][k * NG + g], B[j][n][k], acc[n][ai], 0, 0, 0);
            }
    };
    constexpr std::true_type REC{};
    constexpr std::false_type INP{};
    const int c0 = s > 0 ? 0 : NH;
    tdw_static_for<D - 1>([&](auto jc) { issue(c0 + decltype(jc)::value, jc); });
    ldState();
    TDW_PIN();
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        const bf16x8 wb = wa[f * 64];
#pragma unroll
        for (int n = 0; n < NB; ++n) acc[n][f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wb, ones, acc[n][f], 0, 0, 0);
    }
#pragma unroll 1
    for (int c = c0; c < NT; c += D) {
        tdw_static_for<D>([&](auto jc) {
            constexpr int j = decltype(jc)::value;
            TDW_PIN(); issue(c + j + D - 1, std::integral_constant<int, (j + D - 1) % D>{}); TDW_PIN();
            if (c + j < NT) {
                if (GRU && c + j < NH) mma(REC, jc);
                else mma(INP, jc);
            }
        });
    }
    if constexpr (L0) {
        // first layer: W = Whi + Wlo, x = xhi + xlo in bf16, the three leading products (rnn_layer.h)
        const bf16x8 *wp = wa + (size_t)(4 + KH * NG) * 64;
        for (int kx = 0; kx < kx0; ++kx) {
            bf16x8 xhi[NB], xlo[NB];
#pragma unroll
            for (int n = 0; n < NB; ++n) { xhi[n] = xp[n][(2 * kx) * 64]; xlo[n] = xp[n][(2 * kx + 1) * 64]; }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const bf16x8 whi = wp[(size_t)(2 * (kx * NG + g)) * 64], wlo = wp[(size_t)(2 * (kx * NG + g) + 1) * 64];
#pragma unroll
                for (int n = 0; n < NB; ++n) {
                    acc[n][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, xlo[n], acc[n][g], 0, 0, 0);
                    acc[n][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(whi, xhi[n], acc[n][g], 0, 0, 0);
                    acc[n][g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wlo, xhi[n], acc[n][g], 0, 0, 0);
                }
            }
        }
    }
    constexpr float P2L2E = TD_GATE_SCALE_TANH;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        if (n > 0 && bbs[n] == bbs[n - 1]) continue;      // the repeated last block: already done
        float hv[16];
        float sv[5][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float prev = cst[n][r >> 2][r & 3];
            if constexpr (!GRU) {
                const float e0 = __builtin_amdgcn_exp2f(acc[n][0][r]), e1 = __builtin_amdgcn_exp2f(acc[n][1][r]);
                const float e2 = __builtin_amdgcn_exp2f(acc[n][2][r]), e3 = __builtin_amdgcn_exp2f(acc[n][3][r]);
                const float ig = __builtin_amdgcn_rcpf(1.0f + e0), fg = __builtin_amdgcn_rcpf(1.0f + e1);
                const float t2 = __builtin_amdgcn_rcpf(e2 + 1.0f), og = __builtin_amdgcn_rcpf(1.0f + e3);
                const float gg = __builtin_fmaf(-2.0f, t2, 1.0f);
                const float cv = __builtin_fmaf(fg, prev, ig * gg);
                cst[n][r >> 2][r & 3] = cv;
                float t5 = __builtin_amdgcn_exp2f(P2L2E * cv);
                t5 = __builtin_amdgcn_rcpf(t5 + 1.0f);
                hv[r] = og * __builtin_fmaf(-2.0f, t5, 1.0f);
                if constexpr (SAVE) { sv[0][r] = ig; sv[1][r] = fg; sv[2][r] = gg; sv[3][r] = og; sv[4][r] = cv; }
            } else {
                const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[n][0][r]));
                const float zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[n][1][r]));
                float t2 = __builtin_amdgcn_exp2f(__builtin_fmaf(rg, acc[n][3][r], acc[n][2][r]));
                t2 = __builtin_amdgcn_rcpf(t2 + 1.0f);
                const float ng = __builtin_fmaf(-2.0f, t2, 1.0f);
                const float h = __builtin_fmaf(zg, prev, (1.0f - zg) * ng);
                cst[n][r >> 2][r & 3] = h;
                hv[r] = h;
                if constexpr (SAVE) { sv[0][r] = rg; sv[1][r] = zg; sv[2][r] = ng; sv[3][r] = acc[n][3][r] * (1.0f / P2L2E); sv[4][r] = h; }
            }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) *(f32x4 *)(stp[n] + 4 * q) = cst[n][q];
        bf16x8 h0, h1;
#pragma unroll
        for (int j = 0; j < 8; ++j) { h0[j] = (__bf16)hv[j]; h1[j] = (__bf16)hv[8 + j]; }
        op[n][(2 * u) * 64] = h0;
        op[n][(2 * u + 1) * 64] = h1;
        if constexpr (SAVE) {
            float *svb = p.save + (size_t)dir * p.save_dir_stride + ((((size_t)l * p.nbb + bbs[n]) * nu + u) * 5) * 1024 + lane * 16;
#pragma unroll
            for (int k = 0; k < 5; ++k)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *(f32x4 *)(svb + k * 1024 + 4 * q) = f32x4{sv[k][4 * q], sv[k][4 * q + 1], sv[k][4 * q + 2], sv[k][4 * q + 3]};
        }
    }
}

template <int NB, bool L0, bool GRU, bool SAVE>
static int tdw_launch_steps(const TdLayerParams &p, int kx0, int nu, hipStream_t stream) {
    const int ntile = td_cdiv(p.nbb, NB), tasks = nu * p.dirs * ntile;
    int nw = tasks / 256;                         // waves per workgroup: spread the tasks over the CUs first
    nw = nw < 1 ? 1 : (nw > 4 ? 4 : nw);
    const int ngrp = td_cdiv(ntile, nw);
    for (int s = 0; s < p.L; ++s)
        hipLaunchKernelGGL((td_rnn_step_wide_kernel<NB, L0, GRU, SAVE>), dim3(8 * td_cdiv(nu, 8 / p.dirs) * ngrp), dim3(64 * nw), 0, stream, p, kx0, s, nu);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

template <int NB>
static int tdw_dispatch_steps(int gru, int save, const TdLayerParams &p, bool layer0, int kx0, int nu, hipStream_t stream) {
    if (layer0) {
        if (!gru) return save ? tdw_launch_steps<NB, true, false, true>(p, kx0, nu, stream) : tdw_launch_steps<NB, true, false, false>(p, kx0, nu, stream);
        return save ? tdw_launch_steps<NB, true, true, true>(p, kx0, nu, stream) : tdw_launch_steps<NB, true, true, false>(p, kx0, nu, stream);
    }
    if (!gru) return save ? tdw_launch_steps<NB, false, false, true>(p, kx0, nu, stream) : tdw_launch_steps<NB, false, false, false>(p, kx0, nu, stream);
    return save ? tdw_launch_steps<NB, false, true, true>(p, kx0, nu, stream) : tdw_launch_steps<NB, false, true, false>(p, kx0, nu, stream);
}

// NUW: unit blocks per wave with four waves
template <int NUW>
static int tdw_dispatch(int gru, int save, const TdLayerParams &p, bool layer0, int kx0, hipStream_t stream) {
    // one workgroup per CU at a time: with fewer single-block workgroups than CUs, pairing blocks would idle CUs
    if (p.nbb * p.dirs > 256) return tdw_dispatch2<NUW, 2, 4>(gru, save, p, layer0, kx0, stream);
    // few workgroups (the reference's batch sizes): eight waves, two per SIMD, each with half the unit blocks
    if constexpr (NUW % 2 == 0) {
        if (p.nbb * p.dirs <= TDW_EIGHT_WAVES_MAX_WG) return tdw_dispatch2<NUW / 2, 1, 8>(gru, save, p, layer0, kx0, stream);
    }
    return tdw_dispatch2<NUW, 1, 4>(gru, save, p, layer0, kx0, stream);
}

// nu: padded 32-unit blocks (2, 3, 4, 6, 8, 12, 16, 24 or 32); kx0: layer-0 input k-steps
int td_launch_layer_wide(int gru, int save, int nu, const TdLayerParams &p, bool layer0, int kx0, hipStream_t stream) {
    if (!p.state) {
        td_set_error("wide recurrent layer: no state scratch");
        return TD_ERR_INVALID;
    }
    if (!layer0 && p.in_frags != p.dirs * 2 * nu) {
        td_set_error("wide recurrent layer: %d input fragments, expected %d", p.in_frags, p.dirs * 2 * nu);
        return TD_ERR_INVALID;
    }
    if (td_use_step_launch(nu, p.nbb, p.dirs)) {
        // two batch blocks per wave (a weight fragment feeds two MFMAs) once one block per wave is more than two waves per
        // CU: from there the launch is bound by the CUs' vector-memory ports (1.25 KiB per MFMA), not by MFMA issue
        // (measured, H = 512: batch 512 5.4 ms per layer with one block against 6.5 with two; batch 2048 16.1 against 12.3)
        if (nu * p.dirs * p.nbb > 512) return tdw_dispatch_steps<2>(gru, save, p, layer0, kx0, nu, stream);
        return tdw_dispatch_steps<1>(gru, save, p, layer0, kx0, nu, stream);
    }
    switch (nu) {
        // small batches at H <= 256 (td_use_wide): one unit block per wave, as many waves as blocks
        case 2: return tdw_dispatch2<1, 1, 2>(gru, save, p, layer0, kx0, stream);
        case 3: return tdw_dispatch2<1, 1, 3>(gru, save, p, layer0, kx0, stream);
        case 6: return tdw_dispatch2<1, 1, 6>(gru, save, p, layer0, kx0, stream);
        case 4: return tdw_dispatch<1>(gru, save, p, layer0, kx0, stream);
        case 8: return tdw_dispatch<2>(gru, save, p, layer0, kx0, stream);
        case 12: return tdw_dispatch<3>(gru, save, p, layer0, kx0, stream);
        case 16: return tdw_dispatch<4>(gru, save, p, layer0, kx0, stream);
        case 24: return tdw_dispatch<6>(gru, save, p, layer0, kx0, stream);
        case 32: return tdw_dispatch<8>(gru, save, p, layer0, kx0, stream);
    }
    td_set_error("unsupported hidden block count %d", nu);
    return TD_ERR_UNSUPPORTED;
}

// ---------------------------------------------------------------------------------------------
// head for wide layers: LayerNorm(D*H) -> FiLM -> fc -> (B, C_out, L)   (blocks.py:293-297, :340-341)
// Same arithmetic as td_head_kernel (rnn.hip), but a position's fragments (up to 128 KiB per batch block) do not fit
// one wave's registers: three passes over them (mean, variance, projection; the second and third hit L2), and the
// fc A fragments are built from the fp32 weights on the fly (256 KiB at D*H = 2048: L2-resident, too large for LDS
// next to the per-feature vectors).
// ---------------------------------------------------------------------------------------------
struct TdHeadWideParams {
    const char *act;
    const float *ln_w, *ln_b, *fc_w, *fc_b, *film;
    const int32_t *film_row;
    float *y;
    int film_rows, B, L, nbb, H, kh, dirs, c_out, per_seq;
};

__global__ __launch_bounds__(256) void td_head_wide_kernel(TdHeadWideParams p) {
    extern __shared__ __attribute__((aligned(16))) float hv[];   // three [KT*16] vectors in fragment order (k = 8 hf + j)
    const int KT = p.dirs * p.kh, DH = p.dirs * p.H;
    float *v_a = hv, *v_b = v_a + KT * 16, *v_msk = v_b + KT * 16;
    const float *film0 = p.film + (size_t)(p.film_row ? p.film_row[0] : 0) * 2 * DH;
    for (int idx = threadIdx.x; idx < KT * 16; idx += 256) {
        const int ks = idx / 16, hf = (idx % 16) / 8, j = idx % 8;
        const int d = ks / p.kh, unit = 16 * (ks % p.kh) + 8 * (j >> 2) + 4 * hf + (j & 3);
        const bool ok = unit < p.H;
        const int feat = d * p.H + (ok ? unit : 0);
        const float lw = ok ? p.ln_w[feat] : 0.f, lb = ok ? p.ln_b[feat] : 0.f;
        if (p.per_seq) {
            v_a[idx] = lw; v_b[idx] = lb;
        } else {
            const float sc = ok ? film0[feat] + 1.0f : 0.f, sh = ok ? film0[DH + feat] : 0.f;
            v_a[idx] = lw * sc; v_b[idx] = lb * sc + sh;
        }
        v_msk[idx] = ok ? 1.f : 0.f;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, hf = lane >> 5;
    const size_t ntask = (size_t)p.L * p.nbb;
    for (size_t task = (size_t)blockIdx.x * 4 + wave; task < ntask; task += (size_t)gridDim.x * 4) {
        const int l = (int)(task / p.nbb), bblk = (int)(task % p.nbb);
        const int b = bblk * 32 + m;
        const float *filmb = p.film + (size_t)(b < p.B ? b : 0) * 2 * DH;
        const bf16x8 *fp = (const bf16x8 *)(p.act + task * KT * TD_FRAG_BYTES) + lane;
        const float *va = v_a + hf * 8, *vb = v_b + hf * 8, *vm = v_msk + hf * 8;
        float sum = 0.f;
#pragma unroll 8
        for (int ks = 0; ks < KT; ++ks) {
            const bf16x8 v = fp[ks * 64];
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += (float)v[j];   // padded units hold exact zeros
        }
        sum += __shfl_xor(sum, 32);
        const float mu = sum / (float)DH;
        float sq = 0.f;
#pragma unroll 8
        for (int ks = 0; ks < KT; ++ks) {
            const bf16x8 v = fp[ks * 64];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dlt = ((float)v[j] - mu) * vm[ks * 16 + j];
                sq = fmaf(dlt, dlt, sq);
            }
        }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq / (float)DH + 1e-5f);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // A fragment of fc_w: lane (channel c = lane & 31, hf) holds units 16 ks + 8 (j >> 2) + 4 hf + (j & 3)
        const bool cok = m < p.c_out;
        const float *fcrow = p.fc_w + (size_t)(cok ? m : 0) * DH;
#pragma unroll 2
        for (int ks = 0; ks < KT; ++ks) {
            const bf16x8 v = fp[ks * 64];
            const int d = ks / p.kh, unit0 = 16 * (ks % p.kh) + 4 * hf;
            bf16x8 yh, yl, wh, wl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int unit = unit0 + 8 * (j >> 2) + (j & 3);
                const bool ok = unit < p.H;
                float av = va[ks * 16 + j], bv = vb[ks * 16 + j];
                if (p.per_seq) {
                    const int feat = d * p.H + (ok ? unit : 0);
                    const float sc = filmb[feat] + 1.0f, sh = filmb[DH + feat];
                    bv = (bv * sc + sh) * vm[ks * 16 + j];
                    av = av * sc;
                }
                const float yv = ((float)v[j] - mu) * rstd * av + bv;
                const __bf16 h = (__bf16)yv;
                yh[j] = h;
                yl[j] = (__bf16)(yv - (float)h);
                const float w = (cok && ok) ? fcrow[d * p.H + unit] : 0.f;
                const __bf16 whj = (__bf16)w;
                wh[j] = whj;
                wl[j] = (__bf16)(w - (float)whj);
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, yh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, yl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, yh, acc, 0, 0, 0);
        }
        if (b < p.B) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * hf;
                if (c < p.c_out) p.y[((size_t)b * p.c_out + c) * p.L + l] = acc[r] + p.fc_b[c];
            }
        }
    }
}

int td_rnn_head_wide(const td_rnn_config *cfg, const TdRnnGeom &g, const float *const *misc, const char *act, const float *film,
                     int film_rows, const int32_t *film_row, int batch, int L, float *y, hipStream_t stream) {
    TdHeadWideParams p;
    p.act = act;
    p.ln_w = misc[TD_LN_W]; p.ln_b = misc[TD_LN_B]; p.fc_w = misc[TD_FC_W]; p.fc_b = misc[TD_FC_B];
    p.film = film; p.film_row = film_row; p.film_rows = film_rows;
    p.y = y; p.B = batch; p.L = L; p.nbb = td_cdiv(batch, TD_WAVE_BATCH);
    p.H = cfg->hidden; p.kh = g.kh; p.dirs = g.dirs; p.c_out = cfg->c_out;
    p.per_seq = (!film_row && film_rows > 1) ? 1 : 0;
    const size_t ntask = (size_t)L * p.nbb;
    int blocks = (int)((ntask + 3) / 4);
    if (blocks > 2048) blocks = 2048;
    const size_t shm = (size_t)3 * g.dirs * g.kh * 16 * sizeof(float);
    hipLaunchKernelGGL(td_head_wide_kernel, dim3(blocks), dim3(256), shm, stream, p);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

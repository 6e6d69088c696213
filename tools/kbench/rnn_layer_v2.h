// The persistent recurrent-layer kernel (one launch = one layer, both directions, all time steps).
//
// Decomposition: a workgroup = 4 waves, one wave per SIMD, each wave owns 32 sequences for the
// whole sequence length: its cell state c (fp32) stays in registers, its hidden state h (bf16
// B-fragments) in a private LDS region, so there is no cross-wave exchange of state and no grid
// sync.  Per time step the gate pre-activations
//     gates^T[4H x 32] = [W_hh | W_ih] (A operand, bf16) x [h_{t-1} ; x_t]^T (B operand, bf16)
// are accumulated in fp32 by v_mfma_f32_32x32x16_bf16.  The weights do not fit one CU
// (1.5 MiB per direction at H = 256), so they stream L2 -> LDS as a ring of two 32 KiB stages by
// LDS-DMA (global_load_lds_dwordx4), shared by the four waves; forward and backward directions
// run in different workgroups (even / odd blockIdx, i.e. on different XCDs, so each XCD's L2 holds
// one direction's weights).
#pragma once
#include "../../timediff_amd/csrc/common.h"
#include "../../timediff_amd/csrc/rnn_layout.h"

#include <type_traits>

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

struct TdLayerParams {
    const char *wpacked;  // packed weights of this layer, direction 0
    size_t dir_stride;    // bytes to direction 1
    const char *xin;      // input fragments  [l][bblk][in_frags][64][8] bf16
    char *out;            // output fragments [l][bblk][out_frags][64][8] bf16
    int in_frags, out_frags;
    int L, nbb, dirs;
    // training only (SAVE kernels): gate activations and state for the backward pass,
    // [dir][l][bblk][u][5][64 lanes][16] fp32 in accumulator layout: LSTM i,f,g,o,c ; GRU r,z,n,(W_hn h + b_hn),h
    float *save;
    size_t save_dir_stride;  // floats
};

template <int NU, int KX, bool L0, bool GRU>
struct TdLayerTraits {
    static constexpr int NG = GRU ? 3 : 4;
    static constexpr int KH = 2 * NU;
    static constexpr int XFR = L0 ? 2 * KX : KX;
    static constexpr int FU = 4 + (KH + XFR) * NG;  // 4 bias fragments lead each unit block
    static constexpr int NF = NU * FU;
    static constexpr int NS = (NF + TD_STAGE_FRAGS - 1) / TD_STAGE_FRAGS;
    static constexpr int LDS_W = 2 * TD_STAGE_BYTES;
    static constexpr int LDS_H = TD_BLOCK_WAVES * KH * TD_FRAG_BYTES;
    static constexpr int LDS_TOTAL = LDS_W + LDS_H;
    static constexpr int XPD = L0 ? XFR : (KX < 8 ? KX : 8);  // input fragments in flight per wave
    static constexpr int APD = 8;                              // weight fragments read ahead of their MFMA
};

template <int NU, int KX, bool L0, bool GRU, bool SAVE = false>
__global__ __launch_bounds__(256, 1) void td_rnn_layer_kernel(TdLayerParams p) {
    using T = TdLayerTraits<NU, KX, L0, GRU>;
    constexpr int NG = T::NG, KH = T::KH, XFR = T::XFR, FU = T::FU, NS = T::NS;
    __shared__ __attribute__((aligned(16))) char lds[T::LDS_TOTAL];

    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    const int dir = p.dirs == 2 ? (blockIdx.x & 1) : 0;
    const int tile = p.dirs == 2 ? (blockIdx.x >> 1) : blockIdx.x;
    const int bblk = tile * TD_BLOCK_WAVES + wave;
    const bool active = bblk < p.nbb;
    const int bb = active ? bblk : p.nbb - 1;

    const char *wp = p.wpacked + (size_t)dir * p.dir_stride;
    char *hreg = lds + T::LDS_W + wave * (KH * TD_FRAG_BYTES);
    {
        bf16x8 zero = {};
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) *(bf16x8 *)(hreg + ks * TD_FRAG_BYTES + lane * 16) = zero;
    }

    // one stage = 32 fragments; every wave moves 8 of them, 1 KiB per instruction
    // (scalar base + 32-bit lane offset: the address stays in SGPRs, nothing per-lane to keep live)
    const unsigned lane16 = lane * 16;
    const char *wwave = wp + (size_t)wave * (TD_STAGE_FRAGS / TD_BLOCK_WAVES) * TD_FRAG_BYTES;
    char *lwave = lds + wave * (TD_STAGE_FRAGS / TD_BLOCK_WAVES) * TD_FRAG_BYTES;
    auto issue_stage = [&](const char *wsrc, int stage, int buf) {
        const char *sbase = wsrc + (size_t)stage * TD_STAGE_BYTES;
        char *dbase = lwave + buf * TD_STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < TD_STAGE_FRAGS / TD_BLOCK_WAVES; ++i) {
            // keep each piece's base in SGPRs (saddr + 32-bit voffset form); without the laundering the
            // compiler folds the piece offsets into per-lane 64-bit addresses and hoists all of them
            const char *sp = sbase + i * TD_FRAG_BYTES;
            asm volatile("" : "+s"(sp));
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(sp + lane16),
                                             (__attribute__((address_space(3))) void *)(dbase + i * TD_FRAG_BYTES), 16, 0, 0);
        }
    };

    float state[NU][16];  // LSTM: cell state c; GRU: hidden state h in fp32 (parked, see td_park)
#pragma unroll
    for (int u = 0; u < NU; ++u)
#pragma unroll
        for (int r = 0; r < 16; ++r) state[u][r] = td_park(0.f);

    issue_stage(wwave, 0, 0);
    int bufsel = 1;  // toggled at every stage boundary; the first boundary selects buffer 0

    // B operand of the bias k-step: ones at k = 0, 1 (the bias rides in as bf16 hi + lo columns)
    bf16x8 ones = {};
    if (lane < 32) { ones[0] = (__bf16)1.0f; ones[1] = (__bf16)1.0f; }

    for (int s = 0; s < p.L; ++s) {
        const int l = dir ? p.L - 1 - s : s;
        // input fragments of this step: layer 0 keeps its few in registers; above, every unit block
        // re-streams them from L2 through a small register ring (holding all of them costs 128 VGPRs)
        const bf16x8 *xp = (const bf16x8 *)(p.xin + ((size_t)l * p.nbb + bb) * p.in_frags * TD_FRAG_BYTES) + lane;
        bf16x8 xf[T::XPD];
        if constexpr (L0) {
#pragma unroll
            for (int i = 0; i < XFR; ++i) xf[i] = xp[i * 64];
        }
        bf16x8 hn[KH];
        const bool last_step = s + 1 == p.L;
        char *op = p.out + (((size_t)l * p.nbb + bb) * p.out_frags + dir * KH) * TD_FRAG_BYTES + lane * 16;

        // the per-step base pointer is laundered so that nothing derived from it is loop-invariant:
        // otherwise every stage's DMA base gets hoisted out of the time loop and spilled
        const char *wstep = wwave;
        asm volatile("" : "+s"(wstep));

        td_static_for<NU>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int base = u * FU;
            if constexpr (!L0) {
#pragma unroll
                for (int i = 0; i < T::XPD; ++i) xf[i] = xp[i * 64];
            }
            f32x16 acc[4];
#pragma unroll
            for (int g = 0; g < 4; ++g)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[g][r] = 0.f;

            // fetch weight fragment f of this step's stream (compile-time f after unrolling)
            auto wfrag = [&](int f) -> bf16x8 {
#ifdef TD_ABLATE_DMA
                if (false) {
#else
                if (f % TD_STAGE_FRAGS == 0) {
#endif
                    const int stage = f / TD_STAGE_FRAGS;
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of `stage` have landed
                    __syncthreads();                                   // everyone's have; stage-1 is fully consumed
                    bufsel ^= 1;
                    if (stage + 1 < NS) issue_stage(wstep, stage + 1, bufsel ^ 1);
                    else if (!last_step) issue_stage(wstep, 0, bufsel ^ 1);
                }
                return *(const bf16x8 *)(lds + bufsel * TD_STAGE_BYTES + (f % TD_STAGE_FRAGS) * TD_FRAG_BYTES + lane * 16);
            };
            auto hfrag = [&](int ks) -> bf16x8 { return *(const bf16x8 *)(hreg + ks * TD_FRAG_BYTES + lane * 16); };

            // Weight fragments are read APD ahead of the MFMA that consumes them (LDS latency is ~100+ cycles,
            // an MFMA 32): the ring A[] is statically indexed, so it lives in registers.
            constexpr int APD = T::APD;
            bf16x8 A[APD];
            td_static_for<APD>([&](auto ic) { A[decltype(ic)::value] = wfrag(base + decltype(ic)::value); });
            bf16x8 hb[2];
            hb[0] = hfrag(0);

            td_static_for<FU>([&](auto fc) {
                constexpr int f = decltype(fc)::value;
                const bf16x8 a = A[f % APD];
                if constexpr (f + APD < FU) A[f % APD] = wfrag(base + f + APD);
                // pin the read-ahead: left alone, the scheduler sinks every ds_read next to its MFMA and the
                // wave then pays the full LDS latency per MFMA
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (f < 4) {
                    // bias k-step (LSTM: b_ih + b_hh per gate; GRU: r, z, n-input, n-hidden)
                    acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, ones, acc[f], 0, 0, 0);
                } else if constexpr (f - 4 < KH * NG) {
                    // recurrent part: h_{t-1} fragments from this wave's LDS region, one k-step ahead
                    constexpr int ks = (f - 4) / NG, g = (f - 4) % NG;
                    if constexpr (g == 0 && ks + 1 < KH) hb[(ks + 1) % 2] = hfrag(ks + 1);
                    constexpr int ai = (GRU && g == 2) ? 3 : g;
                    acc[ai] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, hb[ks % 2], acc[ai], 0, 0, 0);
                } else if constexpr (L0) {
                    // first layer: W = Whi + Wlo, x = xhi + xlo in bf16; keep the three leading products
                    constexpr int x = f - 4 - KH * NG;
                    constexpr int part = x & 1, kx = (x >> 1) / NG, g = (x >> 1) % NG;
                    if constexpr (part == 0) {
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[2 * kx + 1], acc[g], 0, 0, 0);
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[2 * kx], acc[g], 0, 0, 0);
                    } else {
                        acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[2 * kx], acc[g], 0, 0, 0);
                    }
                } else {
                    constexpr int x = f - 4 - KH * NG;
                    constexpr int kx = x / NG, g = x % NG;
                    acc[g] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, xf[kx % T::XPD], acc[g], 0, 0, 0);
                    if constexpr (g == NG - 1 && kx + T::XPD < KX) xf[kx % T::XPD] = xp[(kx + T::XPD) * 64];
                }
            });

            // gate nonlinearities; row r of the accumulator = unit 32u + (r&3) + 8(r>>2) + 4hf
            float hv[16];
            float *svp = nullptr;
            if constexpr (SAVE)
                svp = p.save + (size_t)dir * p.save_dir_stride + ((((size_t)l * p.nbb + bb) * NU + u) * 5) * 1024 + lane * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
#ifdef TD_ABLATE_POINTWISE
                {
                    const float cv = 0.5f * td_unpark(state[u][r]) + 0.01f * (acc[0][r] + acc[1][r] + acc[2][r] + acc[3][r]);
                    state[u][r] = td_park(cv);
                    hv[r] = 0.5f * cv;
                    continue;
                }
#endif
                if constexpr (!GRU) {
                    const float ig = td_sigmoid(acc[0][r]), fg = td_sigmoid(acc[1][r]);
                    const float gg = td_tanh(acc[2][r]), og = td_sigmoid(acc[3][r]);
                    const float cv = fg * td_unpark(state[u][r]) + ig * gg;
                    state[u][r] = td_park(cv);
                    hv[r] = og * td_tanh(cv);
                    if constexpr (SAVE) {
                        if (active) {
                            svp[r] = ig; svp[1024 + r] = fg; svp[2048 + r] = gg; svp[3072 + r] = og; svp[4096 + r] = cv;
                        }
                    }
                } else {
                    const float rg = td_sigmoid(acc[0][r]), zg = td_sigmoid(acc[1][r]);
                    const float ng = td_tanh(acc[2][r] + rg * acc[3][r]);
                    const float h = (1.0f - zg) * ng + zg * td_unpark(state[u][r]);
                    state[u][r] = td_park(h);
                    hv[r] = h;
                    if constexpr (SAVE) {
                        if (active) {
                            svp[r] = rg; svp[1024 + r] = zg; svp[2048 + r] = ng; svp[3072 + r] = acc[3][r]; svp[4096 + r] = h;
                        }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                hn[2 * u][j] = (__bf16)hv[j];
                hn[2 * u + 1][j] = (__bf16)hv[8 + j];
            }
                    // h_t goes to HBM for the next layer right away; keeping the consumer here also pins the
            // gate math to this point of the instruction stream (the accumulators die here).
            if (active) {
                *(bf16x8 *)(op + (2 * u) * TD_FRAG_BYTES) = hn[2 * u];
                *(bf16x8 *)(op + (2 * u + 1) * TD_FRAG_BYTES) = hn[2 * u + 1];
            }
            asm volatile("" : "+v"(hn[2 * u]), "+v"(hn[2 * u + 1]));
        });

        // h_t replaces h_{t-1} in this wave's LDS region only now: every unit block above read the old one
#pragma unroll
        for (int ks = 0; ks < KH; ++ks) *(bf16x8 *)(hreg + ks * TD_FRAG_BYTES + lane * 16) = hn[ks];
    }
}

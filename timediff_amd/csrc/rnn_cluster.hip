// Cluster-persistent recurrent layer (forward): the small-batch kernel of the denoiser (blocks.py:289, nn.LSTM / nn.GRU
// with zero initial state), for the reference's own batch sizes (32 for training, 100 for sampling) up to a few hundred
// sequences.
//
// rnn_layer.h gives 32 sequences to one wave and streams the layer's 1.5 MiB of weights through every CU once per
// time step; rnn_wide.hip splits the units over the waves of ONE workgroup.  Either way a batch of 32 sequences keeps
// two CUs of 256 busy and a time step costs the L2 -> CU transfer of the whole weight matrix (21 us at H = 256).  Here
// a batch block of 32 sequences and one direction is worked by a CLUSTER of NU workgroups on NU CUs (NU = H / 32):
//   * workgroup c owns the 32 hidden units [32c, 32c + 32): wave w the 8 units [32c + 8w, +8) with all their gates,
//     as ONE 32-row MFMA tile whose rows are ordered (gate, unit), so that lane (sequence m, half hf) ends up with
//     every gate of its 4 units in its 16 accumulator registers and the gate nonlinearities need no exchange;
//   * the wave's slice of [W_hh | W_ih] (KH + KX fragments of 1 KiB = 192 registers at H = 256, bidirectional input)
//     is loaded ONCE and stays in registers for the whole sequence: nothing is streamed per step but activations;
//   * the input projection of step t + 1 (32 MFMAs, no dependence on the recurrence) runs while h_t is on its way;
//   * per step the NU workgroups exchange h_t (16 KiB) through L2 as 8-byte {payload, tag} granules written and polled
//     with relaxed agent-scope 64-bit atomics (each granule is self-validating: no flag, no fence; MI355X_MICROARCH.md,
//     "Valid forms": 8-byte agent atomics on both sides).  The tag is (launch epoch, step), so stale data of an earlier
//     step or launch never matches.  A workgroup may overwrite a parity buffer only after it has received every
//     peer's h of the step in between, which each peer produced after its own reads of the older buffer.
// Every workgroup of a launch must be resident at once (NU * dirs * batch blocks <= 256 CUs, one workgroup per CU by
// its register use): the launcher checks this; every spin is bounded, and a timeout poisons the output with NaNs
// (a hang would take the GPU with it, a wrong-but-finite result would go unnoticed).
//
// Activations in and out, and the values saved for the backward pass, use the layouts of the other kernel families
// (rnn_layout.h, rnn_params.h): the cluster kernel is a drop-in per layer launch.
#include "rnn_cluster.h"
#include "rnn_params.h"

// packed weights of one (layer, direction): [c (NU)][wave (4)] slices, each = 64 lanes x 16 fp32 bias (4 KiB) followed by
// NFR fragments of 1 KiB (bf16): KH recurrent k-steps, then the input k-steps (layer 0: hi, lo per 16 channels)
__host__ __device__ static inline size_t tdc_slice_bytes(int nfr) { return 4096 + (size_t)nfr * TD_FRAG_BYTES; }

struct TdClusterPack {
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    char *dst;
    int H, In, nu, kh, nfr, layer0, kx0, dirs_in, gru;
};

// gate slot -> torch row block and scale.  LSTM slots i, f, g, o; GRU slots r, z, n (input part), n (hidden part).
__device__ __forceinline__ int tdc_row_block(int gru, int slot) { return gru ? (slot < 2 ? slot : 2) : slot; }
__device__ __forceinline__ float tdc_scale(int gru, int slot) {
    const bool tanh_gate = gru ? slot >= 2 : slot == 2;
    return tanh_gate ? TD_GATE_SCALE_TANH : TD_GATE_SCALE_SIGMOID;
}

#define TDC_PACK_JOBS 12
struct TdClusterPackJobs {
    TdClusterPack job[TDC_PACK_JOBS];
};

__global__ void td_cluster_pack_kernel(TdClusterPackJobs jobs) {   // blockIdx.y: (layer, direction) of the call
    const TdClusterPack &p = jobs.job[blockIdx.y];
    const size_t slice = tdc_slice_bytes(p.nfr);
    const size_t per_slice = 1024 + (size_t)p.nfr * 512;   // elements: 1024 fp32 bias + nfr * 512 bf16
    const size_t total = (size_t)p.nu * TDC_WAVES * per_slice;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int sl = (int)(idx / per_slice);
        const size_t e = idx % per_slice;
        const int c = sl / TDC_WAVES, w = sl % TDC_WAVES;
        char *base = p.dst + (size_t)sl * slice;
        if (e < 1024) {
            // bias as the accumulator's initial value: lane (m, hf), register 4 slot + i <-> unit 32c + 8w + 4hf + i
            const int lane = (int)(e / 16), reg = (int)(e % 16);
            const int hf = lane >> 5, slot = reg >> 2, i = reg & 3;
            const int unit = 32 * c + 8 * w + 4 * hf + i;
            float b = 0.f;
            if (unit < p.H) {
                const int rb = tdc_row_block(p.gru, slot);
                if (!p.gru || slot < 2) b = p.b_ih[rb * p.H + unit] + p.b_hh[rb * p.H + unit];
                else if (slot == 2) b = p.b_ih[2 * p.H + unit];
                else b = p.b_hh[2 * p.H + unit];
            }
            ((float *)base)[e] = b * tdc_scale(p.gru, slot);
            continue;
        }
        const size_t fe = e - 1024;
        const int f = (int)(fe / 512), lane = (int)((fe % 512) / 8), j = (int)(fe % 8);
        const int rho = lane & 31, hf = lane >> 5;          // A operand: lane holds row rho, k = 8 hf + j of the k-step
        const int slot = rho >> 3, unit = 32 * c + 8 * w + (rho & 7);
        const int rb = tdc_row_block(p.gru, slot);
        const float s = tdc_scale(p.gru, slot);
        float v = 0.f;
        bool lo = false;
        if (unit < p.H) {
            if (f < p.kh) {            // recurrent k-step f: hidden unit order of the activation fragments
                const int kk = 16 * f + 8 * (j >> 2) + 4 * hf + (j & 3);
                if (kk < p.H && !(p.gru && slot == 2)) v = p.w_hh[(size_t)(rb * p.H + unit) * p.H + kk];
            } else if (p.layer0) {     // first layer: natural channel order, bf16 hi + lo parts
                const int x = f - p.kh, kx = x >> 1;
                lo = x & 1;
                const int ch = 16 * kx + 8 * hf + j;
                if (ch < p.In && !(p.gru && slot == 3)) v = p.w_ih[(size_t)(rb * p.H + unit) * p.In + ch];
            } else {
                const int x = f - p.kh, din = x / p.kh, ks = x % p.kh;
                const int kk = 16 * ks + 8 * (j >> 2) + 4 * hf + (j & 3);
                if (kk < p.H && !(p.gru && slot == 3)) v = p.w_ih[(size_t)(rb * p.H + unit) * p.In + din * p.H + kk];
            }
        }
        v *= s;
        if (lo) v = v - (float)(__bf16)v;
        ((__bf16 *)(base + 4096))[fe] = (__bf16)v;
    }
}

struct TdClusterParams {
    const char *wpacked;      // this layer, direction 0
    size_t dir_stride;
    const char *xin;          // input fragments  [l][bblk][in_frags][64][8] bf16
    char *out;                // output fragments [l][bblk][out_frags][64][8] bf16
    float *save;              // training: [dir][l][bblk][u][5][64][16] fp32 (rnn_params.h)
    size_t save_dir_stride;
    unsigned long long *xch;  // [dir][bblk][parity][NU * 4 waves * 64 lanes * 2] granules
    float *cstate;            // [dir][bblk][c][wave][64 lanes][4] fp32: recurrent state of clusters that work several batch blocks
    int in_frags, out_frags, L, nbb, dirs;
    int ngroups;              // clusters per direction; cluster gi works the batch blocks gi, gi + ngroups, ...
    unsigned epoch;
    unsigned *err;            // host word raised when a poll times out (api.cpp), or NULL
};

// acc += sum_f W[w0 + f] x (B fragment f of `src`), f < NF: the B fragments come from LDS through a register ring RD deep,
// each read issued RD MFMAs ahead of its use.  Left to the compiler the reads share one register quad and every MFMA waits
// for its own ds_read (150 cycles per MFMA instead of 32); the scheduling barriers pin the interleaving.
#define TDC_RING 8
template <int N, class F>
__device__ __forceinline__ void td_cluster_unrolled(F &&f) {
    if constexpr (N > 0) {
        td_cluster_unrolled<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}
#define TDC_LDS_STATE_BLOCKS 16   // blocks per cluster whose recurrent state fits the LDS left beside the staging buffers
template <int NF, int W0, int NW, int RING = TDC_RING>
__device__ __forceinline__ void tdc_mfma_stream(f32x16 &acc, const bf16x8 (&W)[NW], const bf16x8 *src) {
    constexpr int RD = NF < RING ? NF : RING;
    bf16x8 R[RD];
#pragma unroll
    for (int i = 0; i < RD; ++i) R[i] = src[(size_t)i * 64];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[W0 + f], R[f % RD], acc, 0, 0, 0);
        if (f + RD < NF) R[f % RD] = src[(size_t)(f + RD) * 64];
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Gate nonlinearities of one (step, batch block) for this lane's 4 units, and everything that leaves the wave:
// the recurrent state (to the scratch when the cluster works several blocks), h_t as tagged granules for the cluster,
// h_t as activation fragment for the next layer, the saved gate values for the backward pass.
// PAYLOAD (the kernel for several blocks per cluster): h_t leaves as untagged 8-byte stores into the parity's 16 KiB of
// B-fragment-ordered payload at `xch` (the flag that covers it is raised by the caller behind a barrier).
// HAND: how h_t reaches the units' other owners.  0: tagged granule pairs (one block per cluster), 1: untagged payload
// (several blocks per cluster), 2: `xch` is the LDS buffer of the next step (the whole layer width in ONE workgroup).
template <int NU, bool GRU, bool SAVE, int HAND = 0>
__device__ __forceinline__ void tdc_finish_step(const TdClusterParams &p, const f32x16 &acc, f32x4 &state, bool dead, float *cptr,
                                                unsigned long long *xch, unsigned tag0, int s, int l, int bb, int dir, int c, int w,
                                                int lane) {
    constexpr int KH = 2 * NU;
    // register 4 slot + i = slot's pre-activation of unit i, already scaled for exp2
    float hval[4];
    f32x4 sv[5];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        constexpr float P2L2E = TD_GATE_SCALE_TANH;
        if constexpr (!GRU) {
            const float ig = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[i]));
            const float fg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[4 + i]));
            const float gg = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(acc[8 + i]) + 1.0f), 1.0f);
            const float og = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[12 + i]));
            const float cv = __builtin_fmaf(fg, state[i], ig * gg);
            state[i] = cv;
            const float th = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(P2L2E * cv) + 1.0f), 1.0f);
            hval[i] = og * th;
            if constexpr (SAVE) { sv[0][i] = ig; sv[1][i] = fg; sv[2][i] = gg; sv[3][i] = og; sv[4][i] = cv; }
        } else {
            const float rg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[i]));
            const float zg = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(acc[4 + i]));
            const float t2 = __builtin_amdgcn_exp2f(__builtin_fmaf(rg, acc[12 + i], acc[8 + i]));
            const float ng = __builtin_fmaf(-2.0f, __builtin_amdgcn_rcpf(t2 + 1.0f), 1.0f);
            const float h = __builtin_fmaf(zg, state[i], (1.0f - zg) * ng);
            state[i] = h;
            hval[i] = h;
            if constexpr (SAVE) { sv[0][i] = rg; sv[1][i] = zg; sv[2][i] = ng; sv[3][i] = acc[12 + i] * (1.0f / P2L2E); sv[4][i] = h; }
        }
    }
    if (dead) {
#pragma unroll
        for (int i = 0; i < 4; ++i) hval[i] = __builtin_nanf("");
    }
    typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 bf16x4;
    bf16x4 hb = {(__bf16)hval[0], (__bf16)hval[1], (__bf16)hval[2], (__bf16)hval[3]};
    const unsigned long long hbits = *(const unsigned long long *)&hb;
    constexpr bool PAYLOAD = HAND == 1;
    if constexpr (PAYLOAD) {
        // the payload goes out FIRST: the caller's counted wait lets the stores below stay in flight (TDC_STORES_BEHIND_PAYLOAD)
        if (s + 1 < p.L)
            __hip_atomic_store((unsigned long long *)((char *)xch + (size_t)(2 * c + (w >> 1)) * TD_FRAG_BYTES + lane * 16 + 8 * (w & 1)), hbits,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("" ::: "memory");
    }
    if (cptr) *(f32x4 *)cptr = state;
    if constexpr (HAND == 2) {
        if (s + 1 < p.L) *(unsigned long long *)((char *)xch + (size_t)(2 * c + (w >> 1)) * TD_FRAG_BYTES + lane * 16 + 8 * (w & 1)) = hbits;
    } else if constexpr (PAYLOAD) {
    } else if (s + 1 < p.L) {   // publish: two granules per lane, tag = (epoch, s + 1)
        const unsigned long long *dst = xch + (size_t)(s & 1) * (NU * TDC_WAVES * 128) + (size_t)(c * TDC_WAVES + w) * 128;
        const unsigned tag = tag0 | (unsigned)(s + 1);
        tdc_store_granule_pair(dst, (unsigned)lane * 16, hbits, tag);
    }
    // the layer's output (next layer / head / weight gradients read it after this launch)
    *(unsigned long long *)(p.out + (((size_t)l * p.nbb + bb) * p.out_frags + dir * KH + 2 * c + (w >> 1)) * TD_FRAG_BYTES + lane * 16 +
                            8 * (w & 1)) = hbits;
    if constexpr (SAVE) {
        float *svp = p.save + (size_t)dir * p.save_dir_stride + ((((size_t)l * p.nbb + bb) * NU + c) * 5) * 1024 + lane * 16 + 4 * w;
#pragma unroll
        for (int q = 0; q < 5; ++q) *(f32x4 *)(svp + q * 1024) = sv[q];
    }
}

#ifdef TDC_TRACE   // diagnostic build only (tools/cluster_trace.py): wall-clock stamps (10 ns) of one wave's step phases
__device__ unsigned long long g_tdc_trace[4096 * 8];
#define TDC_TSTAMP(k) do { if (trace_on) g_tdc_trace[(s & 4095) * 8 + (k)] = wall_clock64(); } while (0)
#define TDC_TSTAMP_V(k, v) do { if (trace_on) g_tdc_trace[(s & 4095) * 8 + (k)] = (v); } while (0)
extern "C" __attribute__((visibility("default"))) int td_debug_cluster_trace(unsigned long long *dst, size_t n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tdc_trace), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#else
#define TDC_TSTAMP(k) do { } while (0)
#define TDC_TSTAMP_V(k, v) do { } while (0)
#endif

template <int NU, int XF, bool L0, bool GRU, bool SAVE>
__global__ __launch_bounds__(256, 1) void td_rnn_cluster_kernel(TdClusterParams p) {
    constexpr int KH = 2 * NU;                       // recurrent k-steps
    constexpr int NFR = KH + XF;                     // weight fragments per wave (layer 0: XF = 2 kx0 hi/lo fragments)
    constexpr int XSTAGE = L0 ? 0 : XF;              // input fragments staged in LDS per step
    constexpr int XLOAD = XSTAGE / TDC_WAVES;        // input fragments each wave brings in per step
    static_assert(L0 || XF % TDC_WAVES == 0, "input fragments must split over the waves");
    __shared__ __attribute__((aligned(16))) char lds[KH * TD_FRAG_BYTES + 2 * (XSTAGE ? XSTAGE : 1) * TD_FRAG_BYTES];
    char *hbuf = lds, *xbuf = lds + KH * TD_FRAG_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int c, grp;
    if (!tdc_block_map(blockIdx.x, p.dirs * p.ngroups, NU, &grp, &c)) return;
    const int dir = grp % p.dirs, gi = grp / p.dirs;
    const int nblk = (p.nbb - gi + p.ngroups - 1) / p.ngroups;   // batch blocks of this cluster (>= 1)
#ifdef TDC_TRACE
    const bool trace_on = grp == 0 && c == 1 && tid == 0;
#endif

    // ---- weights: once, into registers --------------------------------------------------------------------
    const char *wsl = p.wpacked + (size_t)dir * p.dir_stride + (size_t)(c * TDC_WAVES + w) * tdc_slice_bytes(NFR);
    f32x16 bias;
    {
        const f32x4 *bp = (const f32x4 *)wsl + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = bp[q];
            bias[4 * q] = v[0]; bias[4 * q + 1] = v[1]; bias[4 * q + 2] = v[2]; bias[4 * q + 3] = v[3];
        }
    }
    bf16x8 W[NFR];
#pragma unroll
    for (int f = 0; f < NFR; ++f) W[f] = *((const bf16x8 *)(wsl + 4096 + (size_t)f * TD_FRAG_BYTES) + lane);

    const unsigned tag0 = p.epoch << 12;
    // LSTM: c; GRU: h (fp32), units 32c + 8w + 4hf + i.  In registers when the cluster works ONE batch block, in the
    // (L2-resident) state scratch between a block's steps otherwise.
    f32x4 state = {0.f, 0.f, 0.f, 0.f};
    bool dead = false;                            // a spin timed out: stop waiting, poison the output

    auto xrow = [&](int l, int bb) { return p.xin + ((size_t)l * p.nbb + bb) * p.in_frags * TD_FRAG_BYTES; };
    // stage the input fragments of (position l, block bb) into xbuf[par]: wave w brings fragments w, w + 4, ...
    bf16x8 xpre[XLOAD ? XLOAD : 1];
    auto x_issue = [&](int l, int bb) {
        if constexpr (!L0) {
            const bf16x8 *xp = (const bf16x8 *)xrow(l, bb) + lane;
#pragma unroll
            for (int i = 0; i < XLOAD; ++i) xpre[i] = xp[(size_t)(w + TDC_WAVES * i) * 64];
        }
    };
    auto x_commit = [&](int par) {
        if constexpr (!L0) {
            bf16x8 *xd = (bf16x8 *)(xbuf + (size_t)par * XSTAGE * TD_FRAG_BYTES) + lane;
#pragma unroll
            for (int i = 0; i < XLOAD; ++i) xd[(size_t)(w + TDC_WAVES * i) * 64] = xpre[i];
        }
    };
    x_issue(dir ? p.L - 1 : 0, gi);
    x_commit(0);

    int it = 0;       // iteration = (step s, block j): LDS staging alternates with its parity
    for (int s = 0; s < p.L; ++s) {
        const int l = dir ? p.L - 1 - s : s;
        for (int j = 0; j < nblk; ++j, ++it) {
            const int bb = gi + j * p.ngroups;
            __syncthreads();     // xbuf[it & 1] is complete; every wave is done with hbuf and with xbuf[(it + 1) & 1]
            TDC_TSTAMP(0);
            // next iteration's inputs: in flight during this one, written to LDS at its end
            const bool more = j + 1 < nblk || s + 1 < p.L;
            if (more) {
                if (j + 1 < nblk) x_issue(l, bb + p.ngroups);
                else x_issue(dir ? l - 1 : l + 1, gi);
            }
            unsigned long long *xch = p.xch + ((size_t)(dir * p.nbb + bb) * 2) * (NU * TDC_WAVES * 128);
            float *cptr = p.cstate + ((((size_t)(dir * p.nbb + bb) * NU + c) * TDC_WAVES + w) * 64 + lane) * 4;
            if (nblk > 1) state = s > 0 ? *(const f32x4 *)cptr : f32x4{0.f, 0.f, 0.f, 0.f};
            // (a) h_{s-1} of this block from every workgroup of the cluster: issue the granule loads now, look at them after
            //     the input MFMAs
            //     (thread (w, lane) takes the granule pair of lane `lane`, wave w of every workgroup n: its 8 bytes of h)
            const __amdgpu_buffer_rsrc_t src = tdc_rsrc(xch + (size_t)((s + 1) & 1) * (NU * TDC_WAVES * 128), NU * TDC_WAVES * 128 * 8);   // parity of step s - 1
            const unsigned goff = (unsigned)(w * 64 + lane) * 16;
            tdc_u32x4 gr[NU];
            if (s > 0) {
#pragma unroll
                for (int n = 0; n < NU; ++n) gr[n] = tdc_load_granule_pair(src, goff + n * (TDC_WAVES * 64 * 16));
            }
            // (b) input projection of this step
            f32x16 acc = bias;
            if constexpr (L0) {
                const bf16x8 *xp = (const bf16x8 *)xrow(l, bb) + lane;
#pragma unroll
                for (int kx = 0; kx < XF / 2; ++kx) {
                    const bf16x8 xh = xp[(size_t)(2 * kx) * 64], xl = xp[(size_t)(2 * kx + 1) * 64];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[KH + 2 * kx], xl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[KH + 2 * kx + 1], xh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[KH + 2 * kx], xh, acc, 0, 0, 0);
                }
            } else {
                const bf16x8 *xs = (const bf16x8 *)(xbuf + (size_t)(it & 1) * XSTAGE * TD_FRAG_BYTES) + lane;
                tdc_mfma_stream<XF, KH, NFR>(acc, W, xs);
            }
            TDC_TSTAMP(1);
            // (c) complete the hand-off: every granule must carry the tag of step s - 1
            if (s > 0) {
                const unsigned want = tag0 | (unsigned)s;      // h_{s-1} was published with tag (epoch, s)
                int spins = 0;
                while (!dead) {
                    bool ok = true;
#pragma unroll
                    for (int n = 0; n < NU; ++n) ok = ok && gr[n].y == want && gr[n].w == want;
                    if (ok) break;
                    if (++spins > TDC_SPIN_LIMIT) { dead = true; TDC_RAISE(p); break; }
                    __builtin_amdgcn_s_sleep(1);
#pragma unroll
                    for (int n = 0; n < NU; ++n)
                        if (gr[n].y != want || gr[n].w != want) gr[n] = tdc_load_granule_pair(src, goff + n * (TDC_WAVES * 64 * 16));
                }
                TDC_TSTAMP(2);
                TDC_TSTAMP_V(6, (unsigned long long)spins);
                // the pair of (workgroup n, wave w, lane)  ->  fragment 2n + (w >> 1), lane, bytes 8 (w & 1) .. + 8
#pragma unroll
                for (int n = 0; n < NU; ++n)
                    *(unsigned long long *)(hbuf + (size_t)(2 * n + (w >> 1)) * TD_FRAG_BYTES + lane * 16 + 8 * (w & 1)) =
                        ((unsigned long long)gr[n].z << 32) | gr[n].x;
            }
            __syncthreads();     // h_{s-1} is in LDS as B fragments
            TDC_TSTAMP(3);
            // (d) recurrent part
            if (s > 0) {
                const bf16x8 *hs = (const bf16x8 *)hbuf + lane;
                tdc_mfma_stream<KH, 0, NFR>(acc, W, hs);
            }
#ifdef TDC_TRACE
            asm volatile("s_nop 0" :: "v"(acc[0]));
#endif
            TDC_TSTAMP(4);
            tdc_finish_step<NU, GRU, SAVE>(p, acc, state, dead, nblk > 1 && s + 1 < p.L ? cptr : nullptr, xch, tag0, s, l, bb, dir, c, w, lane);
            TDC_TSTAMP(5);
            if (more) x_commit((it + 1) & 1);
        }
    }
}

// The same cluster, working SEVERAL batch blocks per step (batches beyond one block per cluster): block j's h_{s-1} was
// published a whole round of blocks ago, so an iteration's operands are brought in during the iteration before it -- by
// LDS-DMA straight into the other half of the double-buffered LDS staging, not through registers: the first version held
// the prefetched granules and input fragments in registers, the compiler had to park them in the accumulator file to make
// room for the MFMA phase, and that move waited for the loads right after issuing them (4 100 of 11 600 cycles per
// iteration).  The hand-off is therefore the bandwidth form (rnn_cluster_bwd.hip): h_t as untagged payload in B-fragment
// order (16 KiB per block and parity, 8-byte write-through stores), one flag word per (block, parity, source workgroup)
// = (launch epoch, step), raised by one lane behind the NEXT iteration's barrier, i.e. after every wave's
// `s_waitcnt vmcnt(0)`; the receiver polls the 8 flags and then lets the DMA (sc1 loads) move the payload.
// A cluster left with ONE block (the batch does not fill the last round) is its own next consumer: it raises the flags
// and stages the next step behind the step's own stores instead.
template <int NU, int XF, bool L0, bool GRU, bool SAVE>
__global__ __launch_bounds__(256, 1) void td_rnn_cluster_multi_kernel(TdClusterParams p) {
    constexpr int KH = 2 * NU;
    constexpr int NFR = KH + XF;
    constexpr int HP = KH / TDC_WAVES;                 // payload pieces each wave moves per iteration
    static_assert(KH % TDC_WAVES == 0, "payload pieces must split over the waves");
    // [h: 2 x KH KiB][x: 2 x XF KiB][recurrent state: a block's 256 threads x 16 B, for up to TDC_LDS_STATE_BLOCKS blocks]
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *hbuf = lds, *xbuf = lds + 2 * KH * TD_FRAG_BYTES;
    f32x4 *state_lds = (f32x4 *)(lds + 2 * (KH + XF) * TD_FRAG_BYTES);

    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int c, grp;
    if (!tdc_block_map(blockIdx.x, p.dirs * p.ngroups, NU, &grp, &c)) return;
    const int dir = grp % p.dirs, gi = grp / p.dirs;
    const int nblk = (p.nbb - gi + p.ngroups - 1) / p.ngroups;

    const char *wsl = p.wpacked + (size_t)dir * p.dir_stride + (size_t)(c * TDC_WAVES + w) * tdc_slice_bytes(NFR);
    f32x16 bias;
    {
        const f32x4 *bp = (const f32x4 *)wsl + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = bp[q];
            bias[4 * q] = v[0]; bias[4 * q + 1] = v[1]; bias[4 * q + 2] = v[2]; bias[4 * q + 3] = v[3];
        }
    }
    bf16x8 W[NFR];
#pragma unroll
    for (int f = 0; f < NFR; ++f) W[f] = *((const bf16x8 *)(wsl + 4096 + (size_t)f * TD_FRAG_BYTES) + lane);
    // The compiler counts vector-memory operations to place its waits and does not see the LDS-DMA pieces issued from inline
    // asm below: with these loads still "pending" at the loop header it waits with small counts inside the MFMA phase, i.e.
    // for the DMA pieces in flight.  Passing the registers through an empty asm makes it wait for them here, once.
#pragma unroll
    for (int f = 0; f < NFR; ++f) asm volatile("" : "+v"(W[f]));
    asm volatile("" : "+v"(bias));

    const unsigned tag0 = p.epoch << 12;
    bool dead = false;
    const bool st_lds = nblk <= TDC_LDS_STATE_BLOCKS;      // the blocks' recurrent state between their steps: LDS, else L2
    auto xrow = [&](int l, int bb) { return p.xin + ((size_t)l * p.nbb + bb) * p.in_frags * TD_FRAG_BYTES; };
    // a block's exchange region (64 KiB): payload of parity 0, of parity 1 (KH KiB each), then [parity][source] flag words
    auto region = [&](int bb) { return (char *)(p.xch + ((size_t)(dir * p.nbb + bb) * 2) * (NU * TDC_WAVES * 128)); };
    auto flags = [&](int bb, int par) { return (unsigned *)(region(bb) + 2 * KH * TD_FRAG_BYTES) + par * NU; };
    auto cstp = [&](int bb) { return p.cstate + ((((size_t)(dir * p.nbb + bb) * NU + c) * TDC_WAVES + w) * 64 + lane) * 4; };

    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)lds;
    // LDS-DMA pieces of 1 KiB (M0 = LDS base of the piece; used by nothing else here); issued from inline asm, covered by
    // the full wait at the top of every iteration
    auto dma = [&](const char *src, unsigned dst) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst), "v"(lane16), "s"(src) : "memory");
    };
    auto dma_sc1 = [&](const char *src, unsigned dst) {
        asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 sc1" ::"s"(dst), "v"(lane16), "s"(src) : "memory");
    };
    // Staging of iteration (step s2 at position l2, block bb2) into the LDS halves `par2`: the wave's pieces are the HP payload
    // pieces of h_{s2-1} (behind the flags) and its share of the XF input fragments.  fpre: the block's flags as read one
    // iteration earlier (a flag read is an L2 round trip, ~2 000 cycles, even when the flag has long been raised).
    constexpr int XP = (XF + TDC_WAVES - 1) / TDC_WAVES, NPIECE = HP + XP;
    auto await_flags = [&](int s2, int bb2, unsigned fpre) {
        if (s2 <= 0) return;
        const unsigned *fl = flags(bb2, (s2 - 1) & 1) + (lane & (NU - 1));
        const unsigned want = tag0 | (unsigned)s2;
        int spins = 0;
        unsigned f = fpre;
        while (!dead) {
            if (__builtin_amdgcn_ballot_w64(f != want) == 0) break;
            if (++spins > TDC_SPIN_LIMIT) { dead = true; TDC_RAISE(p); break; }
            if (spins > 1) __builtin_amdgcn_s_sleep(1);
            f = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        asm volatile("" ::: "memory");      // the payload loads stay behind the poll
    };
    const char *f_h = nullptr, *f_x = nullptr;    // sources of the staging in progress (wave-uniform)
    unsigned f_hd = 0, f_xd = 0;
    bool f_has_h = false;
    auto stage_begin = [&](int s2, int l2, int bb2, int par2) {
        f_has_h = s2 > 0;
        f_h = region(bb2) + (size_t)((s2 + 1) & 1) * KH * TD_FRAG_BYTES + (size_t)w * HP * TD_FRAG_BYTES;
        f_x = xrow(l2, bb2);
        asm volatile("" : "+s"(f_h), "+s"(f_x));
        f_hd = lds0 + (unsigned)((par2 * KH + w * HP) * TD_FRAG_BYTES);
        f_xd = lds0 + (unsigned)((2 * KH + par2 * XF) * TD_FRAG_BYTES);
    };
    auto stage_piece = [&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr (k < HP) {
            if (f_has_h) dma_sc1(f_h + k * TD_FRAG_BYTES, f_hd + k * TD_FRAG_BYTES);
        } else if constexpr (k < NPIECE) {
            constexpr int i = k - HP;
            if (w + TDC_WAVES * i < XF) dma(f_x + (size_t)(w + TDC_WAVES * i) * TD_FRAG_BYTES, f_xd + (w + TDC_WAVES * i) * TD_FRAG_BYTES);
        }
    };
    auto fetch = [&](int s2, int l2, int bb2, int par2, unsigned fpre) {
        await_flags(s2, bb2, fpre);
        stage_begin(s2, l2, bb2, par2);
        td_cluster_unrolled<NPIECE>(stage_piece);
    };
    fetch(0, dir ? p.L - 1 : 0, gi, 0, 0u);
    unsigned fnext = 0;     // flags of the iteration after next, read ahead (0 never equals a tag of this launch: epoch >= 1)
    bool counted = false;   // the iteration before issued exactly TDC_STORES_BEHIND_PAYLOAD stores behind its payload store

    unsigned *pend_flag = nullptr;      // flag of the payload stored in the iteration before, raised behind the next barrier
    unsigned pend_tag = 0;
    f32x4 keep = {0.f, 0.f, 0.f, 0.f};  // a single block's recurrent state stays in registers
    int it = 0;
#ifdef TDC_STAMP
    // diagnostic build (-DTDC_STAMP): cycles per phase of wave 0 of workgroup 0, printed at the end
    unsigned long long stp[6] = {0, 0, 0, 0, 0, 0}, st_last = __builtin_amdgcn_s_memtime();
    const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime(), st_c0 = st_last;
#define TDC_MARK(k) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); stp[k] += now_ - st_last; st_last = now_; }
#else
#define TDC_MARK(k)
#endif
    for (int s = 0; s < p.L; ++s) {
        const int l = dir ? p.L - 1 - s : s;
        for (int j = 0; j < nblk; ++j, ++it) {
            const int bb = gi + j * p.ngroups, par = it & 1;
            TDC_MARK(5)
            // this iteration's h and x have landed in LDS, and every vector-memory operation of this wave so far has completed --
            // the payload stores of the iteration before among them; behind the barrier that holds for every wave, and the
            // other halves of the LDS buffers are free again
            // (the stores issued behind the payload store are younger -- vmcnt retires in order -- and may stay in flight)
            if (counted) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(SAVE ? 6 : 1) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TDC_MARK(0)
            __syncthreads();
            TDC_MARK(1)
            if (pend_tag) {
                if (tid == 0) __hip_atomic_store(pend_flag, pend_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pend_tag = 0;
            }
            f32x4 state = keep;
            if (nblk > 1 && s > 0) state = st_lds ? state_lds[j * 256 + tid] : *(const f32x4 *)cstp(bb);
            const bool last_blk = j + 1 == nblk;
            const int s2 = last_blk ? s + 1 : s, bb2 = last_blk ? gi : bb + p.ngroups;
            const int l2 = dir ? p.L - 1 - s2 : s2;
            const bool staging = nblk > 1 && s2 < p.L;
            if (staging) {
                await_flags(s2, bb2, fnext);
                stage_begin(s2, l2, bb2, par ^ 1);
                // all pieces at once: spread over the input MFMAs (one behind every second MFMA) they stall the MFMA issue for
                // longer than their address processing takes here (measured: 2 590 -> 4 120 cycles for the MFMA phase)
                td_cluster_unrolled<NPIECE>(stage_piece);
                // the flags the iteration after this one will ask for
                const bool last2 = bb2 + p.ngroups >= p.nbb;
                const int s3 = last2 ? s2 + 1 : s2, bb3 = last2 ? gi : bb2 + p.ngroups;
                fnext = (s3 > 0 && s3 < p.L) ? __hip_atomic_load(flags(bb3, (s3 - 1) & 1) + (lane & (NU - 1)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
            }
            TDC_MARK(2)
            // input projection + recurrent part
            f32x16 acc = bias;
            const bf16x8 *xs = (const bf16x8 *)(xbuf + (size_t)par * XF * TD_FRAG_BYTES) + lane;
            if constexpr (L0) {
#pragma unroll
                for (int kx = 0; kx < XF / 2; ++kx) {
                    const bf16x8 xh = xs[(size_t)(2 * kx) * 64], xl = xs[(size_t)(2 * kx + 1) * 64];
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[KH + 2 * kx], xl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[KH + 2 * kx + 1], xh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[KH + 2 * kx], xh, acc, 0, 0, 0);
                }
            } else {
                tdc_mfma_stream<XF, KH, NFR>(acc, W, xs);
            }
            if (s > 0) {
                const bf16x8 *hs = (const bf16x8 *)(hbuf + (size_t)par * KH * TD_FRAG_BYTES) + lane;
                tdc_mfma_stream<KH, 0, NFR>(acc, W, hs);
            }
            TDC_MARK(3)
            // gates, state, payload, outputs
            char *pay = region(bb) + (size_t)(s & 1) * KH * TD_FRAG_BYTES;
            tdc_finish_step<NU, GRU, SAVE, 1>(p, acc, state, dead, nblk > 1 && !st_lds && s + 1 < p.L ? cstp(bb) : nullptr,
                                                 (unsigned long long *)pay, tag0, s, l, bb, dir, c, w, lane);
            if (nblk == 1) keep = state;
            else if (st_lds) state_lds[j * 256 + tid] = state;
            counted = nblk > 1 && st_lds && s + 1 < p.L;      // payload, then the output (+ 5 saved quantities)
            if (s + 1 < p.L) {
                pend_flag = flags(bb, s & 1) + c;
                pend_tag = tag0 | (unsigned)(s + 1);
                if (nblk == 1) {
                    // the only block: the next iteration consumes these very values
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (tid == 0) __hip_atomic_store(pend_flag, pend_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pend_tag = 0;
                    fetch(s2, l2, bb2, par ^ 1, 0u);
                }
            }
            TDC_MARK(4)
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // no LDS-DMA piece in flight when the workgroup's LDS is handed on
#ifdef TDC_STAMP
    if (blockIdx.x == 0 && tid == 0) {
        const unsigned long long cyc = __builtin_amdgcn_s_memtime() - st_c0, rt = __builtin_amdgcn_s_memrealtime() - st_r0;
        printf("cluster multi XF=%d SAVE=%d: %d iterations, %llu cycles (%.0f MHz); per iteration: wait %llu barrier %llu poll+dma %llu mfma %llu finish %llu loop %llu\n",
               XF, (int)SAVE, it, cyc, (double)cyc / (double)rt * 100.0, stp[0] / it, stp[1] / it, stp[2] / it, stp[3] / it, stp[4] / it, stp[5] / it);
    }
#endif
}

// ---------------------------------------------------------------------------------------------
// Widths of up to three unit blocks (H <= 96: the reference's default widths for MIMIC-III, 92, and HiRID, 36,
// etdiff_train.py:239): the "cluster" is ONE workgroup of 4 NU waves on one CU.  The same decomposition -- wave (c, w)
// owns the 8 units [32c + 8w, +8) with all their gates as one 32-row MFMA tile, its slice of [W_hh | W_ih] (at most 18
// fragments = 72 registers) stays in registers for the whole sequence -- but h_t changes hands through LDS (double
// buffered, one barrier per step) instead of through L2, so a step costs the CU's own MFMA and LDS time and no
// round trip.  rnn_wide.hip, which these widths used before, streams the layer's weights from L2 every step (200 KiB at
// H = 92: 3.9 us per step; here about a third of that).  One workgroup per (batch block, direction) and no hand-off
// between workgroups: any batch runs, nothing needs to be co-resident.
template <int NU, int XF, bool L0, bool GRU, bool SAVE>
__global__ __launch_bounds__(256 * NU, 1) void td_rnn_resident_kernel(TdClusterParams p) {
    constexpr int KH = 2 * NU, NFR = KH + XF, NW = TDC_WAVES * NU;
    constexpr int XSTAGE = L0 ? 0 : XF;              // input fragments staged in LDS per step
    constexpr int XLOAD = (XSTAGE + NW - 1) / NW;    // input fragments each wave brings in per step (the last round may be short)
    constexpr int RING = NU == 3 ? 4 : TDC_RING;     // three waves per SIMD: 168 registers each
    __shared__ __attribute__((aligned(16))) char lds[2 * KH * TD_FRAG_BYTES + 2 * (XSTAGE ? XSTAGE : 1) * TD_FRAG_BYTES];
    char *hbuf = lds, *xbuf = lds + 2 * KH * TD_FRAG_BYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = wave >> 2, w = wave & 3;
    const int dir = blockIdx.x % p.dirs, bb = blockIdx.x / p.dirs;

    const char *wsl = p.wpacked + (size_t)dir * p.dir_stride + (size_t)wave * tdc_slice_bytes(NFR);
    f32x16 bias;
    {
        const f32x4 *bp = (const f32x4 *)wsl + lane * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x4 v = bp[q];
            bias[4 * q] = v[0]; bias[4 * q + 1] = v[1]; bias[4 * q + 2] = v[2]; bias[4 * q + 3] = v[3];
        }
    }
    bf16x8 W[NFR];
#pragma unroll
    for (int f = 0; f < NFR; ++f) W[f] = *((const bf16x8 *)(wsl + 4096 + (size_t)f * TD_FRAG_BYTES) + lane);

    f32x4 state = {0.f, 0.f, 0.f, 0.f};      // LSTM: c; GRU: h (fp32) of units 32c + 8w + 4hf + i
    auto xrow = [&](int l) { return p.xin + ((size_t)l * p.nbb + bb) * p.in_frags * TD_FRAG_BYTES; };
    bf16x8 xpre[XLOAD ? XLOAD : 1];
    auto x_issue = [&](int l) {
        if constexpr (!L0) {
            const bf16x8 *xp = (const bf16x8 *)xrow(l) + lane;
#pragma unroll
            for (int i = 0; i < XLOAD; ++i)
                if (wave + NW * i < XSTAGE) xpre[i] = xp[(size_t)(wave + NW * i) * 64];
        }
    };
    auto x_commit = [&](int par) {
        if constexpr (!L0) {
            bf16x8 *xd = (bf16x8 *)(xbuf + (size_t)par * XSTAGE * TD_FRAG_BYTES) + lane;
#pragma unroll
            for (int i = 0; i < XLOAD; ++i)
                if (wave + NW * i < XSTAGE) xd[(size_t)(wave + NW * i) * 64] = xpre[i];
        }
    };
    x_issue(dir ? p.L - 1 : 0);
    x_commit(0);
    // first layer: the few input fragments (hi / lo parts) go straight to registers, one step ahead
    bf16x8 x0[L0 ? XF : 1];
    auto x0_load = [&](int l) {
        if constexpr (L0) {
            const bf16x8 *xp = (const bf16x8 *)xrow(l) + lane;
#pragma unroll
            for (int f = 0; f < XF; ++f) x0[f] = xp[(size_t)f * 64];
        }
    };
    x0_load(dir ? p.L - 1 : 0);

    for (int s = 0; s < p.L; ++s) {
        const int l = dir ? p.L - 1 - s : s;
        __syncthreads();     // h_{s-1} and this step's inputs are in LDS; every wave is done with the other halves
        if (s + 1 < p.L) x_issue(dir ? l - 1 : l + 1);
        f32x16 acc = bias;
        if constexpr (L0) {
#pragma unroll
            for (int kx = 0; kx < XF / 2; ++kx) {
                const bf16x8 xh = x0[2 * kx], xl = x0[2 * kx + 1];
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[KH + 2 * kx], xl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[KH + 2 * kx + 1], xh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[KH + 2 * kx], xh, acc, 0, 0, 0);
            }
            if (s + 1 < p.L) x0_load(dir ? l - 1 : l + 1);
        } else {
            const bf16x8 *xs = (const bf16x8 *)(xbuf + (size_t)(s & 1) * XSTAGE * TD_FRAG_BYTES) + lane;
            tdc_mfma_stream<XF, KH, NFR, RING>(acc, W, xs);
        }
        if (s > 0) {
            const bf16x8 *hs = (const bf16x8 *)(hbuf + (size_t)(s & 1) * KH * TD_FRAG_BYTES) + lane;
            tdc_mfma_stream<KH, 0, NFR, RING>(acc, W, hs);
        }
        tdc_finish_step<NU, GRU, SAVE, 2>(p, acc, state, false, nullptr, (unsigned long long *)(hbuf + (size_t)((s + 1) & 1) * KH * TD_FRAG_BYTES),
                                          0u, s, l, bb, dir, c, w, lane);
        if (s + 1 < p.L) x_commit((s + 1) & 1);
    }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
// Every workgroup of a cluster launch must be resident at once, one per CU: the grid is sized from the device's own count
// of compute units (a partition or a CU mask the runtime reports shrinks it); a device shared with another process cannot
// be seen from here -- there the bounded polls time out and td_check_async_errors reports it.
int td_cluster_device_cus(void) {
    static int cus[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return TDC_CUS;
    if (cus[dev] == 0) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = TDC_CUS;
        cus[dev] = n < TDC_CUS ? n : TDC_CUS;
    }
    return cus[dev];
}

static unsigned g_cluster_epoch = 0;
unsigned td_cluster_next_epoch(void) { return (++g_cluster_epoch) & 0xFFFFFu; }

size_t td_cluster_packed_bytes(const TdRnnGeom &g, int layer0, int kx0) {
    const int nfr = g.kh + (layer0 ? 2 * kx0 : g.dirs * g.kh);
    return (size_t)g.nu * TDC_WAVES * tdc_slice_bytes(nfr);
}

// scratch per direction: per batch block 2 parities x NU x 4 waves x 128 granules of 8 B (exchange), then NU x 4 waves x 64
// lanes x 4 fp32 of recurrent state (all batch blocks' exchange buffers first, then all state)
size_t td_cluster_xch_dir_bytes(const TdRnnGeom &g, int nbb) {
    return (size_t)nbb * (2 * g.nu * TDC_WAVES * 128 * 8 + (size_t)g.nu * TDC_WAVES * 64 * 16);
}

bool td_cluster_supported(const TdRnnGeom &g, int nbb, int L) {
    // tags carry 12 bits of step index; a parity buffer must have been rewritten by this launch before a step index of the
    // previous launch can recur (L >= 4).  Any batch: a cluster works its batch blocks one after the other within a step.
    if (g.nu <= TDC_RESIDENT_MAX_NU) return L >= 1 && nbb >= 1;      // one workgroup per (block, direction): nothing to co-schedule
    return g.nu == 8 && L >= 4 && L < 4096 && nbb >= 1 && td_cluster_device_cus() >= g.nu * g.dirs;
}

int td_cluster_pack_all(const td_rnn_config *cfg, const TdRnnGeom &g, const float *const *lw, char *packed, hipStream_t stream) {
    TdClusterPackJobs jobs;
    int nj = 0;
    for (int l = 0; l < cfg->layers; ++l)
        for (int d = 0; d < cfg->dirs; ++d) {
            const float *const *w = lw + (size_t)(l * cfg->dirs + d) * 4;
            TdClusterPack &p = jobs.job[nj++];
            p.w_ih = w[0]; p.w_hh = w[1]; p.b_ih = w[2]; p.b_hh = w[3];
            p.dst = packed + g.cl_off[l] + (size_t)d * g.cl_dir[l == 0 ? 0 : 1];
            p.H = cfg->hidden; p.In = l == 0 ? cfg->c_in : cfg->dirs * cfg->hidden;
            p.nu = g.nu; p.kh = g.kh; p.layer0 = l == 0; p.kx0 = g.kx0; p.dirs_in = g.dirs; p.gru = cfg->gru;
            p.nfr = g.kh + (l == 0 ? 2 * g.kx0 : g.dirs * g.kh);
            if (nj == TDC_PACK_JOBS || (l == cfg->layers - 1 && d == cfg->dirs - 1)) {
                hipLaunchKernelGGL(td_cluster_pack_kernel, dim3(1024, nj), dim3(256), 0, stream, jobs);
                TD_LAUNCH_CHECK();
                nj = 0;
            }
        }
    return TD_OK;
}

template <int NU, int XF, bool L0>
static int tdc_launch2(int gru, int save, const TdClusterParams &p, int blocks, hipStream_t stream) {
    if (p.nbb > p.ngroups) {   // some cluster works two or more batch blocks: the pipelined kernel
        const int nblk = td_cdiv(p.nbb, p.ngroups);
        const size_t shm = (size_t)2 * (2 * NU + XF) * TD_FRAG_BYTES + (nblk <= TDC_LDS_STATE_BLOCKS ? (size_t)nblk * 256 * 16 : 0);
        auto launch = [&](auto kern) -> int {
            TD_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), shm, stream, p);
            TD_LAUNCH_CHECK();
            return TD_OK;
        };
        if (!gru && !save) return launch(td_rnn_cluster_multi_kernel<NU, XF, L0, false, false>);
        if (!gru) return launch(td_rnn_cluster_multi_kernel<NU, XF, L0, false, true>);
        if (!save) return launch(td_rnn_cluster_multi_kernel<NU, XF, L0, true, false>);
        return launch(td_rnn_cluster_multi_kernel<NU, XF, L0, true, true>);
    }
    if (!gru && !save) hipLaunchKernelGGL((td_rnn_cluster_kernel<NU, XF, L0, false, false>), dim3(blocks), dim3(256), 0, stream, p);
    else if (!gru) hipLaunchKernelGGL((td_rnn_cluster_kernel<NU, XF, L0, false, true>), dim3(blocks), dim3(256), 0, stream, p);
    else if (!save) hipLaunchKernelGGL((td_rnn_cluster_kernel<NU, XF, L0, true, false>), dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((td_rnn_cluster_kernel<NU, XF, L0, true, true>), dim3(blocks), dim3(256), 0, stream, p);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

template <int NU, int XF, bool L0>
static int tdc_launch_resident(int gru, int save, const TdClusterParams &p, hipStream_t stream) {
    const dim3 grid(p.nbb * p.dirs), block(256 * NU);
    if (!gru && !save) hipLaunchKernelGGL((td_rnn_resident_kernel<NU, XF, L0, false, false>), grid, block, 0, stream, p);
    else if (!gru) hipLaunchKernelGGL((td_rnn_resident_kernel<NU, XF, L0, false, true>), grid, block, 0, stream, p);
    else if (!save) hipLaunchKernelGGL((td_rnn_resident_kernel<NU, XF, L0, true, false>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((td_rnn_resident_kernel<NU, XF, L0, true, true>), grid, block, 0, stream, p);
    TD_LAUNCH_CHECK();
    return TD_OK;
}
template <int NU>
static int tdc_launch_resident_nu(int gru, int save, const TdRnnGeom &g, const TdClusterParams &p, bool layer0, hipStream_t stream) {
    if (layer0) return g.kx0 == 1 ? tdc_launch_resident<NU, 2, true>(gru, save, p, stream) : tdc_launch_resident<NU, 4, true>(gru, save, p, stream);
    return p.dirs == 1 ? tdc_launch_resident<NU, 2 * NU, false>(gru, save, p, stream) : tdc_launch_resident<NU, 4 * NU, false>(gru, save, p, stream);
}

// One layer, both directions.  `wcluster`: this layer's cluster-packed weights (direction 0), `dir_stride` to direction 1.
int td_launch_layer_cluster(int gru, int save, const TdRnnGeom &g, const TdLayerParams &lp, const char *wcluster, size_t dir_stride,
                            bool layer0, unsigned long long *xch, hipStream_t stream) {
    TdClusterParams p;
    p.wpacked = wcluster; p.dir_stride = dir_stride;
    p.xin = lp.xin; p.out = lp.out; p.save = lp.save; p.save_dir_stride = lp.save_dir_stride;
    p.xch = xch; p.in_frags = lp.in_frags; p.out_frags = lp.out_frags; p.L = lp.L; p.nbb = lp.nbb; p.dirs = lp.dirs;
    if (g.nu <= TDC_RESIDENT_MAX_NU) {      // the layer's whole width in one workgroup: no exchange buffer, no epoch
        p.cstate = nullptr; p.ngroups = lp.nbb; p.epoch = 0; p.err = nullptr;
        if (g.nu == 1) return tdc_launch_resident_nu<1>(gru, save, g, p, layer0, stream);
        if (g.nu == 2) return tdc_launch_resident_nu<2>(gru, save, g, p, layer0, stream);
        return tdc_launch_resident_nu<3>(gru, save, g, p, layer0, stream);
    }
    // the state scratch follows the exchange buffers of both directions
    p.cstate = (float *)((char *)xch + (size_t)lp.dirs * lp.nbb * 2 * g.nu * TDC_WAVES * 128 * 8);
    // one workgroup per CU, every one resident at once: at most 256 / (NU * dirs) clusters per direction
    const int max_groups = td_cluster_device_cus() / (g.nu * lp.dirs);
    if (max_groups < 1) {
        td_set_error("cluster kernel: the device has fewer than %d compute units", g.nu * lp.dirs);
        return TD_ERR_UNSUPPORTED;
    }
    p.ngroups = lp.nbb < max_groups ? lp.nbb : max_groups;
    p.epoch = td_cluster_next_epoch();
    p.err = td_async_error_word(false);
    const int blocks = tdc_grid(lp.dirs * p.ngroups, g.nu);
    if (g.nu != 8) {
        td_set_error("cluster kernel: hidden block count %d is not instantiated", g.nu);
        return TD_ERR_UNSUPPORTED;
    }
    if (layer0) {
        if (g.kx0 == 1) return tdc_launch2<8, 2, true>(gru, save, p, blocks, stream);
        return tdc_launch2<8, 4, true>(gru, save, p, blocks, stream);
    }
    if (lp.dirs == 1) return tdc_launch2<8, 16, false>(gru, save, p, blocks, stream);
    return tdc_launch2<8, 32, false>(gru, save, p, blocks, stream);
}

// Cluster-persistent BPTT sweep: the backward-in-time loop of one recurrent layer (torch autograd through nn.LSTM /
// nn.GRU, blocks.py:289) for the small batches of the reference's training configuration (32 sequences per micro-batch).
//
// td_bptt_sweep_kernel (rnn_train.hip) gives a batch block and direction to ONE workgroup, which reads the layer's
// W_hh^T (512 KiB at H = 256) from L2 every step: 14 us per step, two CUs of 256 busy.  Here the same cluster as the
// forward pass (rnn_cluster.hip) works it: NU workgroups on NU CUs, workgroup c owning the units [32c, 32c + 32) and
// wave w of it the 8 units [32c + 8w, +8), in the register order of the forward cluster kernel's saved gate values.
//   * The recurrent product d h_{t-1} = W_hh^T d a_t is split along its CONTRACTION index: workgroup c multiplies the
//     d a of its own 32 units (4 gates x 2 k-steps, staged through LDS as B fragments) with its 64 KiB slice of
//     W_hh^T, which stays in registers for the whole sweep (16 fragments per wave: two 32-row blocks of the result
//     each), and sends the 32 x 32 partial sums of row block c' to workgroup c' as bf16.
//   * The partial sums travel through L2 as 8-byte {2 x bf16, tag} granules (rnn_cluster.h), 16 written and 16 polled
//     per lane and step; the receiver adds the 8 partials in fp32.  A parity buffer is rewritten two steps later, which
//     a sender can only reach after every receiver has consumed the step in between (it needs their partial sums,
//     which they send after reading).
//   * d c (LSTM) / z * d h (GRU) carries and the accumulators never leave registers; the saved gate values and the
//     incoming output gradients of the next step are prefetched under the current step's MFMAs.
// d a_t leaves in the two layouts the batched GEMMs after the sweep read (fragments for d x = W_ih^T d a, plain rows for
// d W), exactly as td_bptt_sweep_kernel writes them.
// Batches beyond one block per cluster (NU * dirs * batch blocks > 256 CUs): a cluster works its blocks one after the
// other within a step, up to TDC_BWD_MAX_BLOCKS of them; block j's partial sums were sent a whole round of blocks ago,
// so its granules are fetched one iteration ahead and normally all carry the right tag at the first look; the carries
// of the blocks live in LDS.
#include "rnn_cluster.h"

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 tdc_bf16x4;

// the value of the neighbouring lane (lane ^ 1)
__device__ __forceinline__ unsigned tdc_swap1(unsigned x) { return (unsigned)__builtin_amdgcn_mov_dpp((int)x, 0xB1, 0xF, 0xF, true); }

// MULTI = the cluster works several batch blocks per step.  Then a block's partial sums are needed a whole round of
// blocks after they were sent and the hand-off is built for bandwidth instead of latency (at a full chip the tagged
// 8-byte granules are one fabric write each, 1 M of them per iteration: 7 us): payload without tags, written by 16-byte
// sc1 stores (two neighbouring lanes' 8 bytes each, exchanged by DPP), one flag word per (destination, source) written by
// one wave behind the next iteration's workgroup barrier, after every wave's s_waitcnt vmcnt(0); the receiver polls the
// 8 flags of its block and then reads the payload with 8-byte sc1 loads (MI355X_MICROARCH.md, hand-offs with sc1 loads
// in place of the acquire, first row).
#ifdef TDC_TRACE   // diagnostic build only (tools/cluster_trace.py): wall-clock stamps (10 ns) of one wave's step phases
__device__ unsigned long long g_tdc_btrace[4096 * 8];
#define TDC_TSTAMP(k) do { if (trace_on) g_tdc_btrace[(s & 4095) * 8 + (k)] = wall_clock64(); } while (0)
#define TDC_TSTAMP_V(k, v) do { if (trace_on) g_tdc_btrace[(s & 4095) * 8 + (k)] = (v); } while (0)
extern "C" __attribute__((visibility("default"))) int td_debug_bptt_trace(unsigned long long *dst, size_t n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tdc_btrace), n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#else
#define TDC_TSTAMP(k) do { } while (0)
#define TDC_TSTAMP_V(k, v) do { } while (0)
#endif

template <int NU, bool GRU, bool MULTI>
__global__ __launch_bounds__(256, 1) void td_bptt_cluster_kernel(TdClusterBwdParams p) {
    constexpr int KH = 2 * NU, NT = GRU ? 3 : 4;
    constexpr int RBW = NU / TDC_WAVES;                 // row blocks of the partial product per wave
    constexpr size_t PAR = (size_t)NU * NU * TDC_WAVES * 128;   // granules per parity buffer
    static_assert(NU % TDC_WAVES == 0, "row blocks must split over the waves");
    // d a_t of the 32 own units [parity][slab][half] (16 KiB), then the carries of a cluster that works several batch
    // blocks: [block][thread] f32x4
    extern __shared__ __attribute__((aligned(16))) char lds[];
    f32x4 *carry_lds = (f32x4 *)(lds + 2 * 8 * TD_FRAG_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, hf = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int c, grp;
    if (!tdc_block_map(blockIdx.x, p.dirs * p.ngroups, NU, &grp, &c)) return;
    const int dir = grp % p.dirs, gi0 = grp / p.dirs;
    const int nblk = (p.nbb - gi0 + p.ngroups - 1) / p.ngroups;   // batch blocks gi0, gi0 + ngroups, ... (>= 1)
#ifdef TDC_TRACE
    const bool trace_on = grp == 0 && c == 1 && tid == 0;
#endif

    // ---- this workgroup's slice of W_hh^T: rows = every hidden unit (the wave's two row blocks), k = its own 32 units
    bf16x8 W[RBW][NT][2];
#pragma unroll
    for (int rbi = 0; rbi < RBW; ++rbi)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int half = 0; half < 2; ++half)
                W[rbi][t][half] = *((const bf16x8 *)(p.wt + (size_t)dir * p.wt_dir +
                                                    (((size_t)(RBW * w + rbi) * NT + t) * KH + 2 * c + half) * TD_FRAG_BYTES) + lane);

    const unsigned tag0 = p.epoch << 12;
    const size_t cols = (size_t)p.L * p.nbb * 32;
    constexpr int hpad = 32 * NU;
    const size_t sub = (size_t)lane * 16 + 8 * (w & 1);     // this wave's 8 bytes of a fragment's lane row
    const int kfrag = 2 * c + (w >> 1);                     // the k-step fragment its 4 units per lane belong to

    auto lpos = [&](int s) { return dir ? s : p.L - 1 - s; };            // BPTT step s works forward position l
    auto save_at = [&](int l, int bb) {
        return p.save + (size_t)dir * p.save_dir + ((((size_t)l * p.nbb + bb) * NU + c) * 5) * 1024 + lane * 16 + 4 * w;
    };
    auto inbox_of = [&](int bb) { return p.inbox + (size_t)(dir * p.nbb + bb) * 2 * PAR; };
    // MULTI: [parity][dst][src] flag words of a block, behind the payload of all blocks and directions
    auto flags_of = [&](int bb) { return (unsigned *)(p.inbox + (size_t)p.dirs * p.nbb * 2 * PAR) + (size_t)(dir * p.nbb + bb) * 2 * NU * NU; };
    // the partial sums for this wave's units of (step s, block bb) from every source workgroup n: one granule pair each
    // (re-read below while stale), out of this workgroup's [src][wave][lane] slab of the parity buffer
    tdc_u32x4 gr[NU];
    constexpr unsigned SLAB = NU * TDC_WAVES * 128 * 8;      // bytes per destination workgroup and parity
    const unsigned goff = (unsigned)(w * 64 + lane) * 16;
    auto poll_src = [&](int s, int bb) { return tdc_rsrc(inbox_of(bb) + (size_t)((s - 1) & 1) * PAR + (size_t)c * (SLAB / 8), SLAB); };
    auto poll_issue = [&](__amdgpu_buffer_rsrc_t src) {
#pragma unroll
        for (int n = 0; n < NU; ++n) gr[n] = tdc_load_granule_pair(src, goff + n * (TDC_WAVES * 64 * 16));
    };
    // what an iteration needs from memory, fetched one iteration ahead
    f32x4 sv[5], pv;
    unsigned long long gs[2];
    auto prefetch = [&](int s, int bb) {
        const int l = lpos(s);
        const float *sp = save_at(l, bb);
#pragma unroll
        for (int q = 0; q < 5; ++q) sv[q] = *(const f32x4 *)(sp + q * 1024);
        pv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s + 1 < p.L) pv = *(const f32x4 *)(save_at(dir ? l + 1 : l - 1, bb) + 4 * 1024);   // c_{t-1} (LSTM) / h_{t-1} (GRU)
        const size_t go = (((size_t)l * p.nbb + bb) * p.out_frags + dir * KH + kfrag) * TD_FRAG_BYTES + sub;
        gs[0] = *(const unsigned long long *)(p.gsrc[0] + go);
        gs[1] = p.n_src > 1 ? *(const unsigned long long *)(p.gsrc[1] + go) : 0ull;
    };
    prefetch(0, gi0);

    f32x4 carry = {0.f, 0.f, 0.f, 0.f};
    if (nblk > 1)
        for (int j = 0; j < nblk; ++j) carry_lds[j * 256 + tid] = carry;     // own slots only: no barrier needed
    float dbacc[4][4];           // bias gradients: d a summed over steps and blocks per lane, over the 32 sequences at the end
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int i = 0; i < 4; ++i) dbacc[gi][i] = 0.f;
    bool dead = false;
    unsigned *pend_flags = nullptr;     // MULTI: flags of the payload sent in the iteration before, raised behind the next barrier
    unsigned pend_tag = 0;
    // MULTI, several blocks per cluster: an iteration's partial sums were sent a round of blocks ago, so they are fetched one
    // iteration ahead (under the MFMAs of the iteration before, like the saved gate values), behind flags that were read one
    // iteration earlier still: a flag read or a payload read is an L2 round trip of 1 - 2 us even when the data has long been
    // there, and at the top of the iteration nothing hides it (measured: 3.57 us per iteration with both there).
    unsigned long long pay[NU];
    bool pay_ready = false;             // pay[] holds the partial sums of the iteration that begins
    unsigned fnext = 0;                 // the flags of the iteration after next, as read one iteration ahead
    auto flag_word = [&](int s2, int bb2) { return flags_of(bb2) + (size_t)((s2 - 1) & 1) * NU * NU + c * NU + (lane & (NU - 1)); };
    auto await_flags = [&](int s2, int bb2, unsigned f, bool have) {     // the 8 sources' flags of (step s2 - 1, block bb2): lanes poll one each
        const unsigned *fl = flag_word(s2, bb2);
        const unsigned want = tag0 | (unsigned)s2;
        int spins = 0;
        while (!dead) {
            if (have && __builtin_amdgcn_ballot_w64(f != want) == 0) break;
            if (++spins > TDC_SPIN_LIMIT) { dead = true; TDC_RAISE(p); break; }
            if (spins > 1) __builtin_amdgcn_s_sleep(1);
            f = __hip_atomic_load(fl, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            have = true;
        }
        asm volatile("" ::: "memory");      // the payload loads stay behind the poll
    };
    auto load_payload = [&](int s2, int bb2) {
        // payload [parity][dst][src][wave][lane] x 8 bytes (4 bf16): this lane's 4 units from every source
        const unsigned long long *src = inbox_of(bb2) + (size_t)((s2 - 1) & 1) * (PAR / 2) + ((size_t)(c * NU) * TDC_WAVES + w) * 64 + lane;
#pragma unroll
        for (int n = 0; n < NU; ++n) pay[n] = __hip_atomic_load(src + (size_t)n * (TDC_WAVES * 64), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    };
    int it = 0;                  // iteration = (step s, block j): the LDS staging alternates with its parity
    for (int s = 0; s < p.L; ++s) {
        const int l = lpos(s);
        for (int j = 0; j < nblk; ++j, ++it) {
            const int bb = gi0 + j * p.ngroups, par = it & 1;
            unsigned long long *inbox = inbox_of(bb);
            TDC_TSTAMP(0);
            // the factors of the gate gradients that do not depend on d h_t (from the saved gate values of this step)
            float gf[6][4];
            auto gate_factors = [&]() {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float prev = pv[i];
                    if constexpr (!GRU) {
                        const float ig = sv[0][i], fg = sv[1][i], gg = sv[2][i], og = sv[3][i], cv = sv[4][i];
                        const float tc = td_tanh(cv);
                        gf[0][i] = gg * ig * (1.0f - ig);
                        gf[1][i] = prev * fg * (1.0f - fg);
                        gf[2][i] = ig * (1.0f - gg * gg);
                        gf[3][i] = tc * og * (1.0f - og);
                        gf[4][i] = og * (1.0f - tc * tc);      // d c += d h * this
                        gf[5][i] = fg;                         // carry = d c * f
                    } else {
                        const float rg = sv[0][i], zg = sv[1][i], ng = sv[2][i], hn = sv[3][i];
                        const float an = (1.0f - zg) * (1.0f - ng * ng);
                        gf[0][i] = an * hn * rg * (1.0f - rg);
                        gf[1][i] = (prev - ng) * zg * (1.0f - zg);
                        gf[2][i] = an;
                        gf[3][i] = an * rg;
                        gf[4][i] = 0.f;
                        gf[5][i] = zg;
                    }
                }
            };
            // ---- d h_t: gradient from above + the cluster's partial sums of W_hh^T d a_{t+1} -----------------------------
            f32x4 dh;
            {
                const tdc_bf16x4 a = *(const tdc_bf16x4 *)&gs[0], b = *(const tdc_bf16x4 *)&gs[1];
#pragma unroll
                for (int i = 0; i < 4; ++i) dh[i] = (float)a[i] + (float)b[i];
            }
            if (s > 0 && !MULTI) {
                const __amdgpu_buffer_rsrc_t src = poll_src(s, bb);
                const unsigned want = tag0 | (unsigned)s;          // partial sums of step s - 1 carry the tag (epoch, s)
                poll_issue(src);
                gate_factors();                                     // what needs no d h: under the hand-off's flight
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
                TDC_TSTAMP(1);
                TDC_TSTAMP_V(6, (unsigned long long)spins);
#pragma unroll
                for (int n = 0; n < NU; ++n) {
                    dh[0] += __builtin_bit_cast(float, gr[n].x << 16);
                    dh[1] += __builtin_bit_cast(float, gr[n].x & 0xffff0000u);
                    dh[2] += __builtin_bit_cast(float, gr[n].z << 16);
                    dh[3] += __builtin_bit_cast(float, gr[n].z & 0xffff0000u);
                }
            } else {
                gate_factors();
            }
            if (s > 0 && MULTI) {
                if (!pay_ready) {       // a cluster with one block: its sums were sent in the iteration before
                    await_flags(s, bb, 0u, false);
                    load_payload(s, bb);
                }
#pragma unroll
                for (int n = 0; n < NU; ++n) {
                    const unsigned lo = (unsigned)pay[n], hi = (unsigned)(pay[n] >> 32);
                    dh[0] += __builtin_bit_cast(float, lo << 16);
                    dh[1] += __builtin_bit_cast(float, lo & 0xffff0000u);
                    dh[2] += __builtin_bit_cast(float, hi << 16);
                    dh[3] += __builtin_bit_cast(float, hi & 0xffff0000u);
                }
#ifdef TDC_TRACE
                asm volatile("s_nop 0" :: "v"(dh[0]), "v"(dh[3]));
                TDC_TSTAMP(1);
#endif
            }
            if (nblk > 1) carry = carry_lds[j * 256 + tid];
            // ---- gate gradients of this lane's 4 units: d a = (d h, carry) times the factors -----------------------------
            float da[4][4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float g = dh[i];
                if constexpr (!GRU) {
                    const float dcv = carry[i] + g * gf[4][i];
                    da[0][i] = dcv * gf[0][i];
                    da[1][i] = dcv * gf[1][i];
                    da[2][i] = dcv * gf[2][i];
                    da[3][i] = g * gf[3][i];
                    carry[i] = dcv * gf[5][i];
                } else {
                    const float dht = g + carry[i];
                    da[0][i] = dht * gf[0][i];
                    da[1][i] = dht * gf[1][i];
                    da[2][i] = dht * gf[2][i];
                    da[3][i] = dht * gf[3][i];
                    carry[i] = dht * gf[5][i];
                }
            }
            if (nblk > 1) carry_lds[j * 256 + tid] = carry;
            if (dead) {
#pragma unroll
                for (int gi = 0; gi < 4; ++gi)
#pragma unroll
                    for (int i = 0; i < 4; ++i) da[gi][i] = __builtin_nanf("");
            }
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
#pragma unroll
                for (int i = 0; i < 4; ++i) dbacc[gi][i] += da[gi][i];
            // ---- d a_t: to LDS for the cluster's product, to HBM for the batched GEMMs after the sweep ---------------------------
            // (the HBM copies go out here, ahead of the barrier: issued behind the send they delayed the next step's poll loads,
            // which queue behind them, by more than they cost here -- 3.24 against 3.04 us per step, profiles/round3_cluster_steps.md)
            tdc_bf16x4 dab[4];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                dab[gi] = tdc_bf16x4{(__bf16)da[gi][0], (__bf16)da[gi][1], (__bf16)da[gi][2], (__bf16)da[gi][3]};
                *(unsigned long long *)(lds + (size_t)(par * 8 + gi * 2 + (w >> 1)) * TD_FRAG_BYTES + sub) = *(const unsigned long long *)&dab[gi];
            }
            auto store_da = [&]() {
                char *fbase = p.daf + (size_t)dir * p.daf_dir + (((size_t)l * p.nbb + bb) * 4 * KH + kfrag) * TD_FRAG_BYTES + sub;
                __bf16 *pl = p.dapl + (size_t)dir * p.dapl_dir + (size_t)(32 * c + 8 * w + 4 * hf) * cols + (size_t)l * p.nbb * 32 + bb * 32 + m;
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    *(unsigned long long *)(fbase + (size_t)gi * KH * TD_FRAG_BYTES) = *(const unsigned long long *)&dab[gi];
#pragma unroll
                    for (int i = 0; i < 4; ++i) pl[((size_t)gi * hpad + i) * cols] = dab[gi][i];
                }
            };
            if constexpr (!MULTI) store_da();
            // MULTI: every vector-memory operation of this wave so far has completed, the payload stores of the iteration before
            // among them (they are the youngest in flight: the loads issued before them were consumed above, and this
            // iteration's HBM copies of d a go out behind the barrier so that this wait does not cover them)
            if constexpr (MULTI) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            TDC_TSTAMP(2);
            __syncthreads();     // d a_t of the 32 units is in LDS (the other parity is free again two iterations on)
            TDC_TSTAMP(3);
            const bool last_blk = j + 1 == nblk;
            const int s2 = last_blk ? s + 1 : s, bb2 = last_blk ? gi0 : bb + p.ngroups;
            if constexpr (MULTI) {
                // behind the barrier every wave's payload stores of the iteration before are complete: raise its flags
                if (w == 0 && lane < NU && pend_tag) __hip_atomic_store(pend_flags + lane * NU + c, pend_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                pend_tag = 0;
                store_da();
            }
            // ---- the next iteration's operands: in flight under this one's MFMAs --------------------------------------------------
            if (s2 < p.L) prefetch(s2, bb2);
            if constexpr (MULTI) {
                pay_ready = false;
                if (nblk > 1 && s2 > 0 && s2 < p.L) {
                    await_flags(s2, bb2, fnext, true);
                    load_payload(s2, bb2);
                    pay_ready = true;
                    const bool last2 = bb2 + p.ngroups >= p.nbb;
                    const int s3 = last2 ? s2 + 1 : s2, bb3 = last2 ? gi0 : bb2 + p.ngroups;
                    fnext = (s3 > 0 && s3 < p.L) ? __hip_atomic_load(flag_word(s3, bb3), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                }
            }
            if (s + 1 >= p.L) continue;
            // ---- partial sums of W_hh^T d a_t over this workgroup's 32 units, for every hidden unit ------------------------------
            bf16x8 B[4][2];
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
#pragma unroll
                for (int half = 0; half < 2; ++half)
                    B[gi][half] = *((const bf16x8 *)(lds + (size_t)(par * 8 + gi * 2 + half) * TD_FRAG_BYTES) + lane);
            f32x16 acc[RBW];
#pragma unroll
            for (int rbi = 0; rbi < RBW; ++rbi)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[rbi][r] = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int slab = GRU ? (t == 2 ? 3 : t) : t;     // GRU: W_hn multiplies r * d a_n
#pragma unroll
                for (int half = 0; half < 2; ++half)
#pragma unroll
                    for (int rbi = 0; rbi < RBW; ++rbi)
                        acc[rbi] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[rbi][t][half], B[slab][half], acc[rbi], 0, 0, 0);
            }
#ifdef TDC_TRACE
            asm volatile("s_nop 0" :: "v"(acc[0][0]), "v"(acc[RBW - 1][15]));
#endif
            TDC_TSTAMP(4);
            // ---- send: accumulator registers 4w' .. 4w' + 3 of row block rb belong to wave w' of workgroup rb, same lane ---------
            if constexpr (!MULTI) {
                const unsigned tag = tag0 | (unsigned)(s + 1);
#pragma unroll
                for (int rbi = 0; rbi < RBW; ++rbi) {
                    const unsigned long long *dst = inbox + (size_t)(s & 1) * PAR + ((size_t)((RBW * w + rbi) * NU + c) * TDC_WAVES) * 128;
#pragma unroll
                    for (int wp = 0; wp < TDC_WAVES; ++wp) {
                        const tdc_bf16x4 v = {(__bf16)acc[rbi][4 * wp], (__bf16)acc[rbi][4 * wp + 1], (__bf16)acc[rbi][4 * wp + 2],
                                              (__bf16)acc[rbi][4 * wp + 3]};
                        const unsigned long long bits = *(const unsigned long long *)&v;
                        tdc_store_granule_pair(dst + wp * 128, (unsigned)lane * 16, bits, tag);
                    }
                }
                TDC_TSTAMP(5);
            } else {
                const bool odd = lane & 1;
#pragma unroll
                for (int rbi = 0; rbi < RBW; ++rbi) {
                    // [parity][dst][src][wave w'][lane] x 8 bytes; a lane pair writes 16 bytes: the even lane those of wave w' = 2q,
                    // the odd lane those of w' = 2q + 1 (each gets the other's 8 bytes for that wave by DPP)
                    const char *dst = (const char *)(inbox + (size_t)(s & 1) * (PAR / 2) + ((size_t)((RBW * w + rbi) * NU + c) * TDC_WAVES) * 64);
                    unsigned pc[TDC_WAVES][2];
#pragma unroll
                    for (int wp = 0; wp < TDC_WAVES; ++wp) {
                        const tdc_bf16x4 v = {(__bf16)acc[rbi][4 * wp], (__bf16)acc[rbi][4 * wp + 1], (__bf16)acc[rbi][4 * wp + 2],
                                              (__bf16)acc[rbi][4 * wp + 3]};
                        const unsigned long long bits = *(const unsigned long long *)&v;
                        pc[wp][0] = (unsigned)bits;
                        pc[wp][1] = (unsigned)(bits >> 32);
                    }
#pragma unroll
                    for (int q = 0; q < TDC_WAVES / 2; ++q) {
                        const unsigned g0 = tdc_swap1(odd ? pc[2 * q][0] : pc[2 * q + 1][0]);    // what the neighbour needs of mine
                        const unsigned g1 = tdc_swap1(odd ? pc[2 * q][1] : pc[2 * q + 1][1]);
                        const tdc_u32x4 v = odd ? tdc_u32x4{g0, g1, pc[2 * q + 1][0], pc[2 * q + 1][1]}
                                                : tdc_u32x4{pc[2 * q][0], pc[2 * q][1], g0, g1};
                        tdc_store16_sc1(dst + (size_t)(2 * q) * 512, (unsigned)(lane & ~1) * 8 + (odd ? 512u : 0u), v);
                    }
                }
                TDC_TSTAMP(5);
                pend_flags = flags_of(bb) + (size_t)(s & 1) * NU * NU;
                pend_tag = tag0 | (unsigned)(s + 1);
                if (nblk == 1) {
                    // a cluster of this launch that has ONE block (the batch blocks do not divide evenly): its next iteration
                    // is the consumer of these very sums, so the flags cannot wait for that iteration's barrier
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();
                    if (w == 0 && lane < NU) __hip_atomic_store(pend_flags + lane * NU + c, pend_tag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    pend_tag = 0;
                }
            }
        }
    }
    // ---- bias gradients ---------------------------------------------------------------------------------------------------
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = dbacc[gi][i];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o);
            const int unit = 32 * c + 8 * w + 4 * hf + i;
            if (m == 0 && unit < p.H) {
                float *bi = p.db_ih[dir], *bh = p.db_hh[dir];
                if (!GRU || gi < 2) {
                    atomicAdd(&bi[gi * p.H + unit], t);
                    atomicAdd(&bh[gi * p.H + unit], t);
                } else if (gi == 2) {
                    atomicAdd(&bi[2 * p.H + unit], t);
                } else {
                    atomicAdd(&bh[2 * p.H + unit], t);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// Widths of up to three unit blocks (H <= 96): the sweep of one (batch block, direction) in ONE workgroup of 4 NU waves,
// the counterpart of td_rnn_resident_kernel (rnn_cluster.hip).  W_hh^T (at most 72 KiB) stays in registers for the whole
// sweep, split by (row block c', gate q): wave 4c' + q multiplies gate q's d a of ALL units (KH k-steps from LDS) into the
// 32 x 32 partial sum of d h for the units of block c'.  The gate gradients are owned as in the forward kernel -- wave
// 4c + w has the units [32c + 8w, +8), whose saved gate values it reads in the forward kernel's register order -- so the
// partial sums change hands through LDS: wave 4c' + q writes its accumulator, wave 4c' + w adds up registers 4w .. 4w + 3
// of the NT gates' accumulators.  Two barriers per step (d a complete / partial sums complete), nothing leaves the CU but
// d a in the two layouts the GEMMs after the sweep read.  td_bptt_sweep_kernel, which these widths used before, reads
// W_hh^T from L2 every step: 8 us per step at H = 92.
template <int NU, bool GRU>
__global__ __launch_bounds__(256 * NU, 1) void td_bptt_resident_kernel(TdClusterBwdParams p) {
    constexpr int KH = 2 * NU, NT = GRU ? 3 : 4;
    // [d a: 4 gates x KH fragments][partial sums: 4 NU waves x 4 KiB fp32]
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char *dabuf = lds;
    f32x4 *psum = (f32x4 *)(lds + 4 * KH * TD_FRAG_BYTES);

    const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, hf = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = wave >> 2, w = wave & 3;          // as owner of gate gradients: units 32c + 8w + 4hf + i
    const int dir = blockIdx.x % p.dirs, bb = blockIdx.x / p.dirs;
    const bool mul = !GRU || w < NT;                // as multiplier: row block c, gate term w
    const int slab = GRU ? (w == 2 ? 3 : w) : w;    // GRU: W_hn multiplies r * d a_n

    bf16x8 W[KH];
#pragma unroll
    for (int f = 0; f < KH; ++f)
        W[f] = mul ? *((const bf16x8 *)(p.wt + (size_t)dir * p.wt_dir + (((size_t)c * NT + (mul ? w : 0)) * KH + f) * TD_FRAG_BYTES) + lane) : bf16x8{};

    const size_t cols = (size_t)p.L * p.nbb * 32;
    constexpr int hpad = 32 * NU;
    const size_t sub = (size_t)lane * 16 + 8 * (w & 1);
    const int kfrag = 2 * c + (w >> 1);
    auto lpos = [&](int s) { return dir ? s : p.L - 1 - s; };
    auto save_at = [&](int l) {
        return p.save + (size_t)dir * p.save_dir + ((((size_t)l * p.nbb + bb) * NU + c) * 5) * 1024 + lane * 16 + 4 * w;
    };
    f32x4 sv[5], pv;
    unsigned long long gs[2];
    auto prefetch = [&](int s) {
        const int l = lpos(s);
        const float *sp = save_at(l);
#pragma unroll
        for (int q = 0; q < 5; ++q) sv[q] = *(const f32x4 *)(sp + q * 1024);
        pv = f32x4{0.f, 0.f, 0.f, 0.f};
        if (s + 1 < p.L) pv = *(const f32x4 *)(save_at(dir ? l + 1 : l - 1) + 4 * 1024);   // c_{t-1} (LSTM) / h_{t-1} (GRU)
        const size_t go = (((size_t)l * p.nbb + bb) * p.out_frags + dir * KH + kfrag) * TD_FRAG_BYTES + sub;
        gs[0] = *(const unsigned long long *)(p.gsrc[0] + go);
        gs[1] = p.n_src > 1 ? *(const unsigned long long *)(p.gsrc[1] + go) : 0ull;
    };
    prefetch(0);

    f32x4 carry = {0.f, 0.f, 0.f, 0.f};
    float dbacc[4][4];
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int i = 0; i < 4; ++i) dbacc[gi][i] = 0.f;

    for (int s = 0; s < p.L; ++s) {
        const int l = lpos(s);
        f32x4 dh;
        {
            const tdc_bf16x4 a = *(const tdc_bf16x4 *)&gs[0], b = *(const tdc_bf16x4 *)&gs[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) dh[i] = (float)a[i] + (float)b[i];
        }
        if (s > 0) {      // registers 4w .. 4w + 3 of row block c's accumulators, one per gate term
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const f32x4 v = psum[((size_t)(c * 4 + q) * 4 + w) * 64 + lane];
#pragma unroll
                for (int i = 0; i < 4; ++i) dh[i] += v[i];
            }
        }
        float da[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float g = dh[i], prev = pv[i];
            if constexpr (!GRU) {
                const float ig = sv[0][i], fg = sv[1][i], gg = sv[2][i], og = sv[3][i], cv = sv[4][i];
                const float tc = td_tanh(cv);
                const float dcv = carry[i] + g * (og * (1.0f - tc * tc));
                da[0][i] = dcv * (gg * ig * (1.0f - ig));
                da[1][i] = dcv * (prev * fg * (1.0f - fg));
                da[2][i] = dcv * (ig * (1.0f - gg * gg));
                da[3][i] = g * (tc * og * (1.0f - og));
                carry[i] = dcv * fg;
            } else {
                const float rg = sv[0][i], zg = sv[1][i], ng = sv[2][i], hn = sv[3][i];
                const float dht = g + carry[i];
                const float an = (1.0f - zg) * (1.0f - ng * ng);
                da[0][i] = dht * (an * hn * rg * (1.0f - rg));
                da[1][i] = dht * ((prev - ng) * zg * (1.0f - zg));
                da[2][i] = dht * an;
                da[3][i] = dht * (an * rg);
                carry[i] = dht * zg;
            }
        }
#pragma unroll
        for (int gi = 0; gi < 4; ++gi)
#pragma unroll
            for (int i = 0; i < 4; ++i) dbacc[gi][i] += da[gi][i];
        // d a_t: to LDS for the product (all units' fragments of a gate side by side), to HBM for the GEMMs after the sweep
        {
            char *fbase = p.daf + (size_t)dir * p.daf_dir + (((size_t)l * p.nbb + bb) * 4 * KH + kfrag) * TD_FRAG_BYTES + sub;
            __bf16 *pl = p.dapl + (size_t)dir * p.dapl_dir + (size_t)(32 * c + 8 * w + 4 * hf) * cols + (size_t)l * p.nbb * 32 + bb * 32 + m;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const tdc_bf16x4 v = {(__bf16)da[gi][0], (__bf16)da[gi][1], (__bf16)da[gi][2], (__bf16)da[gi][3]};
                const unsigned long long bits = *(const unsigned long long *)&v;
                *(unsigned long long *)(dabuf + (size_t)(gi * KH + kfrag) * TD_FRAG_BYTES + sub) = bits;
                *(unsigned long long *)(fbase + (size_t)gi * KH * TD_FRAG_BYTES) = bits;
#pragma unroll
                for (int i = 0; i < 4; ++i) pl[((size_t)gi * hpad + i) * cols] = v[i];
            }
        }
        __syncthreads();     // d a_t of every unit is in LDS; every wave has read the partial sums of the step before
        if (s + 1 < p.L) prefetch(s + 1);
        if (s + 1 < p.L && mul) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            const bf16x8 *bsrc = (const bf16x8 *)(dabuf + (size_t)slab * KH * TD_FRAG_BYTES) + lane;
#pragma unroll
            for (int f = 0; f < KH; ++f) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(W[f], bsrc[(size_t)f * 64], acc, 0, 0, 0);
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4)
                psum[((size_t)wave * 4 + g4) * 64 + lane] = f32x4{acc[4 * g4], acc[4 * g4 + 1], acc[4 * g4 + 2], acc[4 * g4 + 3]};
        }
        __syncthreads();     // the partial sums are in LDS; every wave is done with d a_t
    }
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float t = dbacc[gi][i];
#pragma unroll
            for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o);
            const int unit = 32 * c + 8 * w + 4 * hf + i;
            if (m == 0 && unit < p.H) {
                float *bi = p.db_ih[dir], *bh = p.db_hh[dir];
                if (!GRU || gi < 2) {
                    atomicAdd(&bi[gi * p.H + unit], t);
                    atomicAdd(&bh[gi * p.H + unit], t);
                } else if (gi == 2) {
                    atomicAdd(&bi[2 * p.H + unit], t);
                } else {
                    atomicAdd(&bh[2 * p.H + unit], t);
                }
            }
        }
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
#define TDC_BWD_MAX_BLOCKS 16     // batch blocks per cluster (their carries live in LDS: 4 KiB each)

static int tdc_bwd_groups(const TdRnnGeom &g, int nbb) {
    const int max_groups = td_cluster_device_cus() / (g.nu * g.dirs);      // every workgroup resident at once, one per CU
    return nbb < max_groups ? nbb : (max_groups > 1 ? max_groups : 1);
}

bool td_bptt_cluster_supported(const TdRnnGeom &g, int nbb, int L) {
    if (g.nu <= TDC_RESIDENT_MAX_NU) return L >= 1 && nbb >= 1;      // one workgroup per (block, direction)
    if (!(g.nu == TD_CLUSTER_NU && L >= 4 && L < 4096 && nbb >= 1 && td_cluster_device_cus() >= g.nu * g.dirs)) return false;
    return td_cdiv(nbb, tdc_bwd_groups(g, nbb)) <= TDC_BWD_MAX_BLOCKS;
}

// inbox bytes for a whole layer launch (both directions); 0 where the kernel cannot run
size_t td_bptt_cluster_inbox_bytes(const TdRnnGeom &g, int nbb) {
    if (g.nu != TD_CLUSTER_NU || td_cdiv(nbb, tdc_bwd_groups(g, nbb)) > TDC_BWD_MAX_BLOCKS) return 0;
    // per (direction, block): 2 parities of granules (or of untagged payload, half the size), then 2 x NU x NU flag words
    return (size_t)g.dirs * nbb * 2 * g.nu * g.nu * (TDC_WAVES * 128 * sizeof(unsigned long long) + sizeof(unsigned));
}

template <bool GRU, bool MULTI>
static int tdc_bwd_launch(const TdClusterBwdParams &p, int blocks, size_t shm, hipStream_t stream) {
    TD_HIP(hipFuncSetAttribute((const void *)td_bptt_cluster_kernel<8, GRU, MULTI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((td_bptt_cluster_kernel<8, GRU, MULTI>), dim3(blocks), dim3(256), shm, stream, p);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

template <int NU, bool GRU>
static int tdc_bwd_launch_resident(const TdClusterBwdParams &p, hipStream_t stream) {
    const size_t shm = (size_t)4 * 2 * NU * TD_FRAG_BYTES + (size_t)4 * NU * 4096;
    TD_HIP(hipFuncSetAttribute((const void *)td_bptt_resident_kernel<NU, GRU>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((td_bptt_resident_kernel<NU, GRU>), dim3(p.nbb * p.dirs), dim3(256 * NU), shm, stream, p);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

int td_launch_bptt_cluster(int gru, const TdRnnGeom &g, TdClusterBwdParams &p, hipStream_t stream) {
    if (g.nu <= TDC_RESIDENT_MAX_NU) {
        p.ngroups = p.nbb; p.epoch = 0; p.err = nullptr;
        if (g.nu == 1) return gru ? tdc_bwd_launch_resident<1, true>(p, stream) : tdc_bwd_launch_resident<1, false>(p, stream);
        if (g.nu == 2) return gru ? tdc_bwd_launch_resident<2, true>(p, stream) : tdc_bwd_launch_resident<2, false>(p, stream);
        return gru ? tdc_bwd_launch_resident<3, true>(p, stream) : tdc_bwd_launch_resident<3, false>(p, stream);
    }
    if (!td_bptt_cluster_supported(g, p.nbb, p.L)) {
        td_set_error("cluster BPTT kernel: %d unit blocks x %d directions x %d batch blocks, L = %d is outside its range", g.nu, g.dirs,
                     p.nbb, p.L);
        return TD_ERR_UNSUPPORTED;
    }
    p.ngroups = tdc_bwd_groups(g, p.nbb);
    p.epoch = td_cluster_next_epoch();
    p.err = td_async_error_word(false);
    const int blocks = tdc_grid(p.dirs * p.ngroups, g.nu);
    const int nblk = td_cdiv(p.nbb, p.ngroups);
    const size_t shm = 2 * 8 * TD_FRAG_BYTES + (nblk > 1 ? (size_t)nblk * 256 * 16 : 0);
    if (nblk > 1) return gru ? tdc_bwd_launch<true, true>(p, blocks, shm, stream) : tdc_bwd_launch<false, true>(p, blocks, shm, stream);
    return gru ? tdc_bwd_launch<true, false>(p, blocks, shm, stream) : tdc_bwd_launch<false, false>(p, blocks, shm, stream);
}

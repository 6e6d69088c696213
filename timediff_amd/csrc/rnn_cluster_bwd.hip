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
// d W), exactly as td_bptt_sweep_kernel writes them.  One batch block per cluster: NU * dirs * batch blocks <= 256.
#include "rnn_cluster.h"

typedef __attribute__((__vector_size__(4 * sizeof(__bf16)))) __bf16 tdc_bf16x4;

template <int NU, bool GRU>
__global__ __launch_bounds__(256, 1) void td_bptt_cluster_kernel(TdClusterBwdParams p) {
    constexpr int KH = 2 * NU, NT = GRU ? 3 : 4;
    constexpr int RBW = NU / TDC_WAVES;                 // row blocks of the partial product per wave
    constexpr size_t PAR = (size_t)NU * NU * TDC_WAVES * 128;   // granules per parity buffer
    static_assert(NU % TDC_WAVES == 0, "row blocks must split over the waves");
    __shared__ __attribute__((aligned(16))) char lds[2 * 8 * TD_FRAG_BYTES];   // d a_t of the 32 own units: [parity][slab][half]

    const int tid = threadIdx.x, lane = tid & 63, m = lane & 31, hf = lane >> 5;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    int c, grp;
    if (!tdc_block_map(blockIdx.x, p.dirs * p.nbb, NU, &grp, &c)) return;
    const int dir = grp % p.dirs, bb = grp / p.dirs;

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
    unsigned long long *inbox = p.inbox + (size_t)(dir * p.nbb + bb) * 2 * PAR;
    const size_t cols = (size_t)p.L * p.nbb * 32;
    constexpr int hpad = 32 * NU;
    const size_t sub = (size_t)lane * 16 + 8 * (w & 1);     // this wave's 8 bytes of a fragment's lane row
    const int kfrag = 2 * c + (w >> 1);                     // the k-step fragment its 4 units per lane belong to

    auto lpos = [&](int s) { return dir ? s : p.L - 1 - s; };            // BPTT step s works forward position l
    auto save_at = [&](int l) {
        return p.save + (size_t)dir * p.save_dir + ((((size_t)l * p.nbb + bb) * NU + c) * 5) * 1024 + lane * 16 + 4 * w;
    };
    // what step s needs from memory, fetched one step ahead
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
    float dbacc[4][4];           // bias gradients: d a summed over the steps per lane, over the 32 sequences at the end
#pragma unroll
    for (int gi = 0; gi < 4; ++gi)
#pragma unroll
        for (int i = 0; i < 4; ++i) dbacc[gi][i] = 0.f;
    bool dead = false;
    for (int s = 0; s < p.L; ++s) {
        const int l = lpos(s), par = s & 1;
        // ---- d h_t: gradient from above + the cluster's partial sums of W_hh^T d a_{t+1} ---------------------------------
        f32x4 dh;
        {
            const tdc_bf16x4 a = *(const tdc_bf16x4 *)&gs[0], b = *(const tdc_bf16x4 *)&gs[1];
#pragma unroll
            for (int i = 0; i < 4; ++i) dh[i] = (float)a[i] + (float)b[i];
        }
        if (s > 0) {
            const unsigned long long *src = inbox + (size_t)((s - 1) & 1) * PAR + ((size_t)(c * NU) * TDC_WAVES + w) * 128 + lane * 2;
            const unsigned want = tag0 | (unsigned)s;          // partial sums of step s - 1 carry the tag (epoch, s)
            unsigned long long gr[2 * NU];
#pragma unroll
            for (int n = 0; n < 2 * NU; ++n)
                gr[n] = __hip_atomic_load(src + (size_t)(n >> 1) * (TDC_WAVES * 128) + (n & 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int spins = 0;
            while (!dead) {
                bool ok = true;
#pragma unroll
                for (int n = 0; n < 2 * NU; ++n) ok = ok && (unsigned)(gr[n] >> 32) == want;
                if (ok) break;
                if (++spins > TDC_SPIN_LIMIT) { dead = true; break; }
                __builtin_amdgcn_s_sleep(1);
#pragma unroll
                for (int n = 0; n < 2 * NU; ++n)
                    if ((unsigned)(gr[n] >> 32) != want)
                        gr[n] = __hip_atomic_load(src + (size_t)(n >> 1) * (TDC_WAVES * 128) + (n & 1), __ATOMIC_RELAXED,
                                                  __HIP_MEMORY_SCOPE_AGENT);
            }
#pragma unroll
            for (int n = 0; n < 2 * NU; ++n) {
                const unsigned pl = (unsigned)gr[n];
                dh[2 * (n & 1)] += __builtin_bit_cast(float, pl << 16);
                dh[2 * (n & 1) + 1] += __builtin_bit_cast(float, pl & 0xffff0000u);
            }
        }
        // ---- gate gradients of this lane's 4 units ------------------------------------------------------------------------
        float da[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float g = dh[i], prev = pv[i];
            if constexpr (!GRU) {
                const float ig = sv[0][i], fg = sv[1][i], gg = sv[2][i], og = sv[3][i], cv = sv[4][i];
                const float tc = td_tanh(cv);
                const float dcv = carry[i] + g * og * (1.0f - tc * tc);
                da[0][i] = dcv * gg * ig * (1.0f - ig);
                da[1][i] = dcv * prev * fg * (1.0f - fg);
                da[2][i] = dcv * ig * (1.0f - gg * gg);
                da[3][i] = g * tc * og * (1.0f - og);
                carry[i] = dcv * fg;
            } else {
                const float rg = sv[0][i], zg = sv[1][i], ng = sv[2][i], hn = sv[3][i];
                const float dht = g + carry[i];
                const float dan = dht * (1.0f - zg) * (1.0f - ng * ng);
                da[0][i] = dan * hn * rg * (1.0f - rg);
                da[1][i] = dht * (prev - ng) * zg * (1.0f - zg);
                da[2][i] = dan;
                da[3][i] = dan * rg;
                carry[i] = dht * zg;
            }
        }
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
        // ---- d a_t: to LDS for the cluster's product, to HBM for the batched GEMMs after the sweep -----------------------------
        {
            char *fbase = p.daf + (size_t)dir * p.daf_dir + (((size_t)l * p.nbb + bb) * 4 * KH + kfrag) * TD_FRAG_BYTES + sub;
            __bf16 *pl = p.dapl + (size_t)dir * p.dapl_dir + (size_t)(32 * c + 8 * w + 4 * hf) * cols + (size_t)l * p.nbb * 32 + bb * 32 + m;
#pragma unroll
            for (int gi = 0; gi < 4; ++gi) {
                const tdc_bf16x4 v = {(__bf16)da[gi][0], (__bf16)da[gi][1], (__bf16)da[gi][2], (__bf16)da[gi][3]};
                const unsigned long long bits = *(const unsigned long long *)&v;
                *(unsigned long long *)(lds + (size_t)(par * 8 + gi * 2 + (w >> 1)) * TD_FRAG_BYTES + sub) = bits;
                *(unsigned long long *)(fbase + (size_t)gi * KH * TD_FRAG_BYTES) = bits;
#pragma unroll
                for (int i = 0; i < 4; ++i) pl[((size_t)gi * hpad + i) * cols] = v[i];
            }
        }
        __syncthreads();     // d a_t of the 32 units is in LDS (the other parity is free again two steps on)
        if (s + 1 >= p.L) break;
        prefetch(s + 1);
        // ---- partial sums of W_hh^T d a_t over this workgroup's 32 units, for every hidden unit ----------------------------------
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
        // ---- send: accumulator registers 4w' .. 4w' + 3 of row block rb belong to wave w' of workgroup rb, same lane -------------
        {
            const unsigned tag = tag0 | (unsigned)(s + 1);
#pragma unroll
            for (int rbi = 0; rbi < RBW; ++rbi) {
                unsigned long long *dst = inbox + (size_t)par * PAR + ((size_t)((RBW * w + rbi) * NU + c) * TDC_WAVES) * 128 + lane * 2;
#pragma unroll
                for (int wp = 0; wp < TDC_WAVES; ++wp) {
                    const tdc_bf16x4 v = {(__bf16)acc[rbi][4 * wp], (__bf16)acc[rbi][4 * wp + 1], (__bf16)acc[rbi][4 * wp + 2],
                                          (__bf16)acc[rbi][4 * wp + 3]};
                    const unsigned long long bits = *(const unsigned long long *)&v;
                    __hip_atomic_store(dst + wp * 128, tdc_granule((unsigned)bits, tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    __hip_atomic_store(dst + wp * 128 + 1, tdc_granule((unsigned)(bits >> 32), tag), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
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
// host side
// ---------------------------------------------------------------------------------------------
// one batch block per cluster, every workgroup resident at once
bool td_bptt_cluster_supported(const TdRnnGeom &g, int nbb, int L) {
    return g.nu == TD_CLUSTER_NU && L >= 4 && L < 4096 && nbb >= 1 && g.nu * g.dirs * nbb <= TDC_CUS;
}

// inbox bytes for a whole layer launch (both directions); 0 where the kernel cannot run
size_t td_bptt_cluster_inbox_bytes(const TdRnnGeom &g, int nbb) {
    if (g.nu != TD_CLUSTER_NU || g.nu * g.dirs * nbb > TDC_CUS) return 0;
    return (size_t)g.dirs * nbb * 2 * g.nu * g.nu * TDC_WAVES * 128 * sizeof(unsigned long long);
}

int td_launch_bptt_cluster(int gru, const TdRnnGeom &g, TdClusterBwdParams &p, hipStream_t stream) {
    if (!td_bptt_cluster_supported(g, p.nbb, p.L)) {
        td_set_error("cluster BPTT kernel: %d unit blocks x %d directions x %d batch blocks, L = %d is outside its range", g.nu, g.dirs,
                     p.nbb, p.L);
        return TD_ERR_UNSUPPORTED;
    }
    p.epoch = td_cluster_next_epoch();
    const int blocks = tdc_grid(p.dirs * p.nbb, g.nu);
    if (gru) hipLaunchKernelGGL((td_bptt_cluster_kernel<8, true>), dim3(blocks), dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((td_bptt_cluster_kernel<8, false>), dim3(blocks), dim3(256), 0, stream, p);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

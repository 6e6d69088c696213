// Mixed-diffusion arithmetic as fused elementwise passes.
// Reference: models/ETDiff/mixed_diffusion.py; n_cat = 0 covers models/ETDiff/gaussian_diffusion.py.
// All of these kernels are HBM-bound streaming passes over channel-major (B, C, L) tensors.  A thread owns VEC = 4
// consecutive positions l of one sequence for ALL channels (16-byte loads and stores along l, one 32-bit load for
// four uint8 class ids); the scalar form VEC = 1 takes sequence lengths that are not a multiple of 4 or pointers
// that are not 16-byte aligned.  Everything that depends only on the step t (and on a variable's class count) is
// computed once per block (reverse update: every sequence is at the same t) or once per thread and variable
// (training: t differs per sequence) instead of once per class and position.  Exponentials and logarithms use the
// hardware forms v_exp_f32 / v_log_f32 (1 ulp) with the base change as one multiply: the deviation from libm's
// expf / logf is a few 1e-7 absolute on every quantity here, against the 1e-5 tolerance of the parity tests; the
// Gumbel transform keeps libm's logf (its inner logarithm's RELATIVE error becomes an absolute error of the noise).
#include "common.h"
#include "rng.h"
#include "rnn_layout.h"

#define TD_KMAX 8          // classes per discrete variable on the vectorised path (logits of 4 positions in registers)
#define TD_KMAX_WIDE 32    // any class count the denoiser width allows (c_out <= 32): scalar path

static int td_check_diff(const td_diffusion_config *c, int batch) {
    TD_REQUIRE(c != nullptr, "td_diffusion_config is NULL");
    TD_REQUIRE(batch > 0, "batch must be positive");
    TD_REQUIRE(c->n_num >= 0 && c->n_num <= 2 * TD_MAX_CAT, "n_num = %d out of range", c->n_num);
    TD_REQUIRE(c->n_cat >= 0 && c->n_cat <= TD_MAX_CAT, "n_cat = %d out of range", c->n_cat);
    TD_REQUIRE(c->n_num + c->n_cat > 0, "no channels");
    TD_REQUIRE(c->seq_len > 0 && c->timesteps > 0, "seq_len and timesteps must be positive");
    TD_REQUIRE(c->tables != nullptr, "tables is NULL");
    for (int v = 0; v < c->n_cat; ++v)
        if (c->num_classes[v] < 1 || c->num_classes[v] > TD_MAX_CLASSES) {
            td_set_error("num_classes[%d] = %d not in [1, %d]", v, c->num_classes[v], TD_MAX_CLASSES);
            return TD_ERR_UNSUPPORTED;
        }
    TD_REQUIRE(c->data_channels >= 1, "data_channels must be positive");
    for (int i = 0; i < c->n_num; ++i)
        TD_REQUIRE(c->num_index[i] >= 0 && c->num_index[i] < c->data_channels, "num_index[%d] out of range", i);
    for (int i = 0; i < c->n_cat; ++i)
        TD_REQUIRE(c->cat_index[i] >= 0 && c->cat_index[i] < c->data_channels, "cat_index[%d] out of range", i);
    return TD_OK;
}

static inline int td_sum_k(const td_diffusion_config *c) {
    int s = 0;
    for (int v = 0; v < c->n_cat; ++v) s += c->num_classes[v];
    return s;
}
static inline int td_max_k(const td_diffusion_config *c) {
    int m = 0;
    for (int v = 0; v < c->n_cat; ++v) m = c->num_classes[v] > m ? c->num_classes[v] : m;
    return m;
}
static inline bool td_al16(const void *p) { return ((uintptr_t)p & 15) == 0; }

__device__ __forceinline__ float td_tab(const td_diffusion_config &c, int row, int t) {
    return c.tables[(size_t)row * c.timesteps + t];
}

// hardware exp / log (v_exp_f32, v_log_f32 are base 2)
__device__ __forceinline__ float td_fexp(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float td_flog(float x) { return __builtin_amdgcn_logf(x) * 0.6931471805599453f; }
// log_add_exp (mixed_diffusion.py:20-22): m + log(exp(a - m) + exp(b - m)); the larger argument's term is exp(0) = 1
__device__ __forceinline__ float td_flae(float a, float b) {
    const float m = fmaxf(a, b);
    return m + td_flog(1.0f + td_fexp(fminf(a, b) - m));
}
// Gumbel transform (:322-324).  Hardware logarithms: the inner one's relative error (2 ulp) becomes an absolute error of
// 2.4e-7 of the noise, the outer one adds 2 ulp of |g| <= 17: below 1e-6 wherever it matters, against the 1e-5 margin gate
// of the parity tests.  TD_GUMBEL_FAST = 0 builds libm's logf for both (A/B measurements: 0.189 -> 0.166 ms per launch).
#ifndef TD_GUMBEL_FAST
#define TD_GUMBEL_FAST 2
#endif
__device__ __forceinline__ float td_gumbel(float u) {
#if TD_GUMBEL_FAST == 0
    return -logf(-logf(u + 1e-30f) + 1e-30f);
#elif TD_GUMBEL_FAST == 1
    return -td_flog(-logf(u + 1e-30f) + 1e-30f);
#else
    return -td_flog(-td_flog(u + 1e-30f) + 1e-30f);
#endif
}

// VEC consecutive floats / class ids along l
template <int VEC>
struct TdVec {
    float v[VEC];
};
template <int VEC>
__device__ __forceinline__ TdVec<VEC> td_ldf(const float *p) {
    TdVec<VEC> r;
    if constexpr (VEC == 4) {
        const f32x4 q = *(const f32x4 *)p;
        r.v[0] = q[0]; r.v[1] = q[1]; r.v[2] = q[2]; r.v[3] = q[3];
    } else {
        r.v[0] = p[0];
    }
    return r;
}
template <int VEC>
__device__ __forceinline__ void td_stf(float *p, const TdVec<VEC> &r) {
    if constexpr (VEC == 4) {
        f32x4 q = {r.v[0], r.v[1], r.v[2], r.v[3]};
        *(f32x4 *)p = q;
    } else {
        p[0] = r.v[0];
    }
}
template <int VEC>
__device__ __forceinline__ void td_ldb(const uint8_t *p, int out[VEC]) {
    if constexpr (VEC == 4) {
        const uint32_t q = *(const uint32_t *)p;
        out[0] = q & 255; out[1] = (q >> 8) & 255; out[2] = (q >> 16) & 255; out[3] = q >> 24;
    } else {
        out[0] = p[0];
    }
}
template <int VEC>
__device__ __forceinline__ void td_stb(uint8_t *p, const int in[VEC]) {
    if constexpr (VEC == 4) *(uint32_t *)p = (uint32_t)in[0] | ((uint32_t)in[1] << 8) | ((uint32_t)in[2] << 16) | ((uint32_t)in[3] << 24);
    else p[0] = (uint8_t)in[0];
}

// Noise source of VEC consecutive positions of one sequence: external arrays or Philox (the counter layout is keyed
// by (patient, position, tag, 4-channel block), so each position makes its own calls whatever VEC is).
template <int VEC>
struct TdDraw {
    const float *noise, *uniforms;
    uint64_t seed;
    uint32_t patient, l0, tag;
    size_t nbase, ubase;  // offsets of (b, channel 0, l0)
    int L;
    float nbuf[VEC][4], ubuf[VEC][TD_UNI_PER_BLOCK];
    int nblk, ublk;
    __device__ TdDraw(const float *n, const float *u, td_rng rng, int b, int l0_, int n_num, int sumk, int L_, uint32_t tag_)
        : noise(n), uniforms(u), seed(rng.seed + (rng.seed_dev ? *rng.seed_dev : 0ull)), patient(rng.patient0 + (uint32_t)b), l0((uint32_t)l0_), tag(tag_),
          nbase(((size_t)b * n_num) * L_ + l0_), ubase(((size_t)b * sumk) * L_ + l0_), L(L_), nblk(-1), ublk(-1) {}
    __device__ TdVec<VEC> normal(int ch) {
        if (noise) return td_ldf<VEC>(noise + nbase + (size_t)ch * L);
        if ((ch >> 2) != nblk) {
            nblk = ch >> 2;
#pragma unroll
            for (int e = 0; e < VEC; ++e) td_normal4(seed, patient, l0 + e, tag, (uint32_t)nblk, nbuf[e]);
        }
        const int q = ch & 3;
        TdVec<VEC> r;
#pragma unroll
        for (int e = 0; e < VEC; ++e) r.v[e] = q == 0 ? nbuf[e][0] : q == 1 ? nbuf[e][1] : q == 2 ? nbuf[e][2] : nbuf[e][3];
        return r;
    }
    __device__ TdVec<VEC> uniform(int slot) {
        if (uniforms) return td_ldf<VEC>(uniforms + ubase + (size_t)slot * L);
        const int blk = slot / TD_UNI_PER_BLOCK, q = slot % TD_UNI_PER_BLOCK;
        if (blk != ublk) {
            ublk = blk;
#pragma unroll
            for (int e = 0; e < VEC; ++e) td_uniform5(seed, patient, l0 + e, tag, (uint32_t)ublk, ubuf[e]);
        }
        TdVec<VEC> r;
#pragma unroll
        for (int e = 0; e < VEC; ++e)
            r.v[e] = q == 0 ? ubuf[e][0] : q == 1 ? ubuf[e][1] : q == 2 ? ubuf[e][2] : q == 3 ? ubuf[e][3] : ubuf[e][4];
        return r;
    }
};

#ifndef TD_GRID_CAP
#define TD_GRID_CAP 65535
#endif
static inline int td_grid(size_t total, int cap = TD_GRID_CAP) {
    size_t b = (total + 255) / 256;
    return (int)(b > (size_t)cap ? cap : (b ? b : 1));
}

// ---------------------------------------------------------------------------------------------
// forward noising for a training batch (mixed_diffusion.py:403-409, :375-387)
// ---------------------------------------------------------------------------------------------
template <int VEC>
__global__ __launch_bounds__(256) void td_q_sample_kernel(td_diffusion_config c, int B, int sumk, const float *__restrict__ data,
                                                          const int64_t *__restrict__ t, const float *noise, const float *uniforms,
                                                          td_rng rng, float *__restrict__ x_in, uint8_t *__restrict__ x0_idx,
                                                          uint8_t *__restrict__ xt_idx, float *__restrict__ noise_out) {
    const int L = c.seq_len, C_io = c.n_num + sumk, LQ = L / VEC;
    const size_t total = (size_t)B * LQ;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / LQ), l0 = (int)(i % LQ) * VEC;
        const int tb = (int)t[b];
        TdDraw<VEC> draw(noise, uniforms, rng, b, l0, c.n_num, sumk, L, rng.tag);
        const float sa = td_tab(c, TD_TAB_SQRT_AC, tb), s1 = td_tab(c, TD_TAB_SQRT_1M_AC, tb);
        for (int ch = 0; ch < c.n_num; ++ch) {
            const TdVec<VEC> d = td_ldf<VEC>(data + ((size_t)b * c.data_channels + c.num_index[ch]) * L + l0);
            const TdVec<VEC> eps = draw.normal(ch);
            TdVec<VEC> o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) o.v[e] = sa * (c.raw_range ? d.v[e] : d.v[e] * 2.0f - 1.0f) + s1 * eps.v[e];  // :189-194, :240-248
            td_stf<VEC>(x_in + ((size_t)b * C_io + ch) * L + l0, o);
            if (noise_out) td_stf<VEC>(noise_out + ((size_t)b * c.n_num + ch) * L + l0, eps);
        }
        const float lca = td_tab(c, TD_TAB_LOG_CUM_ALPHA, tb), l1mca = td_tab(c, TD_TAB_LOG_1M_CUM_ALPHA, tb);
        int off = 0, lastK = -1;
        float lg_hit = 0.f, lg_miss = 0.f;
        for (int v = 0; v < c.n_cat; ++v) {
            const int K = c.num_classes[v];
            if (K != lastK) {   // q(x_t | x_0) takes two values per class count (:264-272): the class of x_0 or not
                lastK = K;
                const float uni = l1mca - logf((float)K);
                lg_hit = td_flae(0.f + lca, uni);
                lg_miss = td_flae(TD_LOG_1E30 + lca, uni);
            }
            const TdVec<VEC> dv = td_ldf<VEC>(data + ((size_t)b * c.data_channels + c.cat_index[v]) * L + l0);
            int x0[VEC], best[VEC];
            float bv[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) { x0[e] = (int)dv.v[e]; best[e] = 0; bv[e] = -INFINITY; }
            for (int k = 0; k < K; ++k) {
                const TdVec<VEC> u = draw.uniform(off + k);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float val = td_gumbel(u.v[e]) + (k == x0[e] ? lg_hit : lg_miss);  // :322-324
                    if (val > bv[e]) { bv[e] = val; best[e] = k; }
                }
            }
            td_stb<VEC>(x0_idx + ((size_t)b * c.n_cat + v) * L + l0, x0);
            td_stb<VEC>(xt_idx + ((size_t)b * c.n_cat + v) * L + l0, best);
            for (int k = 0; k < K; ++k) {
                TdVec<VEC> o;
#pragma unroll
                for (int e = 0; e < VEC; ++e) o.v[e] = k == best[e] ? 0.f : TD_LOG_1E30;
                td_stf<VEC>(x_in + ((size_t)b * C_io + c.n_num + off + k) * L + l0, o);
            }
            off += K;
        }
    }
}

extern "C" int td_q_sample_mixed(const td_diffusion_config *cfg, int32_t batch, const float *data, const int64_t *t,
                                 const float *noise, const float *uniforms, td_rng rng, float *x_in, uint8_t *x0_idx,
                                 uint8_t *xt_idx, float *noise_out, void *stream) {
    int rc = td_check_diff(cfg, batch);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(data && t && x_in, "NULL argument");
    TD_REQUIRE(cfg->n_cat == 0 || (x0_idx && xt_idx), "x0_idx / xt_idx required when n_cat > 0");
    const bool vec = cfg->seq_len % 4 == 0 && td_al16(data) && td_al16(noise) && td_al16(uniforms) && td_al16(x_in) &&
                     td_al16(noise_out) && ((uintptr_t)x0_idx & 3) == 0 && ((uintptr_t)xt_idx & 3) == 0;
    const int sumk = td_sum_k(cfg);
    if (vec)
        hipLaunchKernelGGL(td_q_sample_kernel<4>, dim3(td_grid((size_t)batch * cfg->seq_len / 4)), dim3(256), 0, (hipStream_t)stream,
                           *cfg, batch, sumk, data, t, noise, uniforms, rng, x_in, x0_idx, xt_idx, noise_out);
    else
        hipLaunchKernelGGL(td_q_sample_kernel<1>, dim3(td_grid((size_t)batch * cfg->seq_len)), dim3(256), 0, (hipStream_t)stream,
                           *cfg, batch, sumk, data, t, noise, uniforms, rng, x_in, x0_idx, xt_idx, noise_out);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

// ---------------------------------------------------------------------------------------------
// reverse update (mixed_diffusion.py:412-423, :307-311)
// ---------------------------------------------------------------------------------------------
struct TdRevParams {
    td_diffusion_config c;
    int B, sumk, t_host;
    const float *mo;
    float *z;
    uint8_t *idx;
    const int32_t *t_dev;
    int32_t *t_next;          // NULL, or where one thread leaves t - 1 for the NEXT step's kernels (the sampler's chain alternates
                              // between two words, so no kernel of this step reads the one written here)
    const float *noise, *uniforms;
    size_t noise_stride, uni_stride;
    td_rng rng;
    // TERMS kernels only (parity): the posterior terms of the step; nothing is sampled or updated
    float *x0_out, *mean_out, *log_theta_out;
    // FRAGS kernels only (the sampler's chain): the NEXT step's layer-0 input fragments, nbb batch blocks, kx0 k-tiles
    char *frags;
    int nbb, kx0;
};

// FRAGS forms (the template argument is P, the number of threads that share one item): one workgroup owns 32 sequences x
// TL = 32 / P positions, keeps the updated state of its tile in LDS (4 bytes per continuous channel, ONE byte per discrete
// variable: the one-hot log values are rebuilt from the class id) and writes the tile's layer-0 fragments as whole 1 KiB
// runs, bit for bit what td_state_to_frags_kernel makes of the stored state.  P = 1: a thread updates every channel of its
// four positions (34-45 KiB of LDS at the reference's datasets: three workgroups per CU).  P = 4, for grids that leave
// most CUs idle (batch 256): wave p of the workgroup takes every fourth block of four continuous channels and every fourth
// discrete variable of the SAME 32 sequences x 8 positions, which cuts the chain of dependent loads, Philox calls and
// logarithms a thread walks through by the same factor (the draws are keyed by (sequence, position, channel block), so who
// makes them changes nothing).
static inline size_t td_rf_shm(const td_diffusion_config *c, int kx0, int P) {
    const int TL = 32 / P;
    return (size_t)c->n_num * 32 * (TL + 1) * sizeof(float) + (size_t)c->n_cat * 32 * (TL + 4) + (size_t)16 * kx0 * sizeof(int);
}

// three waves per SIMD (138 registers, no spills).  Measured A/B in one process at B = 16 384, L = 272 (0.164-0.172 ms per
// launch = 2.5-2.7 TB/s): 3 vs 4 waves per SIMD (the latter spills 60 B per lane) and a capped vs one-item-per-thread grid
// are within the run-to-run noise of each other; what holds the kernel is its integer and transcendental work (3 Philox4x32-10
// calls, 2 Box-Muller pairs, 30 logs / exps per position ~ 0.1 ms of VALU issue at this size), not HBM.
#ifndef TD_REV_WAVES
#define TD_REV_WAVES 3
#endif
#ifndef TD_REV_SPLIT_BELOW
#define TD_REV_SPLIT_BELOW 256    // tiles of 32 sequences x 32 positions
#endif
template <int VEC, int KM, bool TERMS, int FRAGS>
__global__ __launch_bounds__(256, TD_REV_WAVES) void td_reverse_step_kernel(TdRevParams p) {
    static_assert(FRAGS == 0 || ((FRAGS == 1 || FRAGS == 4) && VEC == 4 && !TERMS), "the fragment-writing forms are vectorised sampling kernels");
    constexpr int P = FRAGS ? FRAGS : 1, TL = 32 / P, RS = TL + 1, IRS = TL + 4;
    const td_diffusion_config &c = p.c;
    extern __shared__ float rf_tile[];   // FRAGS: z tile [n_num][32][RS], class ids [n_cat][32][IRS], channel codes [16 kx0]
    float *ztile = rf_tile;
    uint8_t *itile = (uint8_t *)(rf_tile + (size_t)c.n_num * 32 * RS);
    int *chan = (int *)(itile + (size_t)c.n_cat * 32 * IRS);
    // per variable: {log K, log q(x_t | x_{t-1}) for the current class, for another class, log((1 - abar_{t-1}) / K)}
    __shared__ float s_var[TD_MAX_CAT][4];
    const int L = c.seq_len, C_io = c.n_num + p.sumk, T = c.timesteps, LQ = L / VEC;
    const int t = p.t_dev ? p.t_dev[0] : p.t_host;
    if (t < 0 || t >= T) return;
    if (p.t_next && blockIdx.x == 0 && threadIdx.x == 0) p.t_next[0] = t - 1;
    const float *noise = p.noise, *uniforms = p.uniforms;
    // chained external draws: slab 0 is the prior, slab T - t belongs to step t
    if (noise && p.noise_stride) noise += (size_t)(T - t) * p.noise_stride;
    if (uniforms && p.uni_stride) uniforms += (size_t)(T - t) * p.uni_stride;
    const float c_recip = td_tab(c, TD_TAB_SQRT_RECIP_AC, t), c_recipm1 = td_tab(c, TD_TAB_SQRT_RECIPM1_AC, t);
    const float c1 = td_tab(c, TD_TAB_POST_C1, t), c2 = td_tab(c, TD_TAB_POST_C2, t);
    const float sigma = expf(0.5f * td_tab(c, TD_TAB_POST_LOGVAR, t));
    const int tm1 = t > 0 ? t - 1 : 0;
    const float lca = td_tab(c, TD_TAB_LOG_CUM_ALPHA, tm1), l1mca = td_tab(c, TD_TAB_LOG_1M_CUM_ALPHA, tm1);
    if ((int)threadIdx.x < c.n_cat) {
        const float la = td_tab(c, TD_TAB_LOG_ALPHA, t), l1ma = td_tab(c, TD_TAB_LOG_1M_ALPHA, t);
        const float logK = logf((float)c.num_classes[threadIdx.x]);
        s_var[threadIdx.x][0] = logK;
        s_var[threadIdx.x][1] = td_flae(0.f + la, l1ma - logK);            // :255-262
        s_var[threadIdx.x][2] = td_flae(TD_LOG_1E30 + la, l1ma - logK);
        s_var[threadIdx.x][3] = l1mca - logK;
    }
    __syncthreads();
    // one (sequence, VEC positions) item, or thread `part`'s share of it; trow / irow = its slot in the LDS tiles (FRAGS)
    auto item = [&](const int b, const int l0, const int trow, const int irow, const int part) {
        TdDraw<VEC> draw(noise, uniforms, p.rng, b, l0, c.n_num, p.sumk, L, (uint32_t)t);
        for (int ch = 0; ch < c.n_num; ++ch) {
            if (P > 1 && ((ch >> 2) & (P - 1)) != part) continue;
            const size_t zi = ((size_t)b * c.n_num + ch) * L + l0;
            const TdVec<VEC> eps = td_ldf<VEC>(p.mo + ((size_t)b * C_io + ch) * L + l0), zt = td_ldf<VEC>(p.z + zi);
            TdVec<VEC> x0, mean, o;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                x0.v[e] = fminf(fmaxf(c_recip * zt.v[e] - c_recipm1 * eps.v[e], -1.0f), 1.0f);   // :196-201, :234-235
                mean.v[e] = c1 * x0.v[e] + c2 * zt.v[e];                                          // :210-218
            }
            if constexpr (TERMS) {
                td_stf<VEC>(p.x0_out + zi, x0);
                td_stf<VEC>(p.mean_out + zi, mean);
            } else {
                if (t > 0) {                                                                       // :421-422
                    const TdVec<VEC> nz = draw.normal(ch);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o.v[e] = mean.v[e] + sigma * nz.v[e];
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) o.v[e] = mean.v[e] + sigma * 0.f;
                }
                td_stf<VEC>(p.z + zi, o);
                if constexpr (FRAGS) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) ztile[ch * 32 * RS + trow + e] = o.v[e];
                }
            }
        }
        int off = 0;
        for (int v = 0; v < c.n_cat; ++v) {
            const int K = c.num_classes[v];
            if (P > 1 && (v & (P - 1)) != P - 1 - part) { off += K; continue; }   // variables from the last wave down, channel blocks from the first up
            const float w_hit = s_var[v][1], w_miss = s_var[v][2], uni = s_var[v][3];
            const float *zp = p.mo + ((size_t)b * C_io + c.n_num + off) * L + l0;
            float logit[KM][VEC], m[VEC], se[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) { m[e] = -INFINITY; se[e] = 0.f; }
#pragma unroll
            for (int k = 0; k < KM; ++k) {
                if (k < K) {
                    const TdVec<VEC> q = td_ldf<VEC>(zp + (size_t)k * L);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { logit[k][e] = q.v[e]; m[e] = fmaxf(m[e], q.v[e]); }
                }
            }
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) se[e] += td_fexp(logit[k][e] - m[e]);
                }
            int xt[VEC], best[VEC];
            float bv[VEC], lse[VEC];
            td_ldb<VEC>(p.idx + ((size_t)b * c.n_cat + v) * L + l0, xt);
#pragma unroll
            for (int e = 0; e < VEC; ++e) { lse[e] = m[e] + td_flog(se[e]); best[e] = 0; bv[e] = -INFINITY; }   // :274-281
            // log pi + log w per class (the per-slice normaliser of :293 shifts all K values equally: argmax-invariant)
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float lx0 = logit[k][e] - lse[e];
                        const float ev = t == 0 ? lx0 : td_flae(lx0 + lca, uni);                  // :284-290
                        logit[k][e] = ev + (k == xt[e] ? w_hit : w_miss);
                    }
                }
            if constexpr (TERMS) {
                float mm[VEC], ss[VEC];
#pragma unroll
                for (int e = 0; e < VEC; ++e) { mm[e] = -INFINITY; ss[e] = 0.f; }
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) mm[e] = fmaxf(mm[e], logit[k][e]);
                    }
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) ss[e] += td_fexp(logit[k][e] - mm[e]);
                    }
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
                        TdVec<VEC> o;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) o.v[e] = logit[k][e] - (mm[e] + td_flog(ss[e]));   // :291-293, direct per-slice LSE
                        td_stf<VEC>(p.log_theta_out + ((size_t)b * p.sumk + off + k) * L + l0, o);
                    }
            } else {
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
                        const TdVec<VEC> u = draw.uniform(off + k);
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const float val = td_gumbel(u.v[e]) + logit[k][e];                    // :322-324
                            if (val > bv[e]) { bv[e] = val; best[e] = k; }
                        }
                    }
                td_stb<VEC>(p.idx + ((size_t)b * c.n_cat + v) * L + l0, best);
                if constexpr (FRAGS) td_stb<VEC>(itile + v * 32 * IRS + irow, best);
            }
            off += K;
        }
    };
    if constexpr (!FRAGS) {
        const size_t total = (size_t)p.B * LQ;
        for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x)
            item((int)(i / LQ), (int)(i % LQ) * VEC, 0, 0, 0);
    } else {
        const int tid = threadIdx.x, bblk = blockIdx.y, lt0 = blockIdx.x * TL;
        const int CP = 16 * p.kx0;
        // channel of the denoiser input -> where its value comes from: a z row, (variable, class), or padding
        for (int ch = tid; ch < CP; ch += 256) {
            int code = -1;
            if (ch < c.n_num) code = ch;
            else if (ch < C_io) {
                int v = 0, o = c.n_num;
                while (ch >= o + c.num_classes[v]) { o += c.num_classes[v]; ++v; }
                code = 0x10000 | (v << 8) | (ch - o);
            }
            chan[ch] = code;
        }
        {
            constexpr int Q = TL / 4;                           // quads of positions per sequence in the tile
            const int part = tid / (32 * Q), r = tid % (32 * Q), bb = r / Q, ll0 = (r % Q) * 4;
            const int b = bblk * 32 + bb, l0 = lt0 + ll0;
            const int trow = bb * RS + ll0, irow = bb * IRS + ll0;
            if (b < p.B && l0 < L) item(b, l0, trow, irow, part);
            else {   // rows past the batch pack as zeros (positions past L are never written)
                for (int ch = 0; ch < c.n_num; ++ch)
                    if (((ch >> 2) & (P - 1)) == part) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) ztile[ch * 32 * RS + trow + e] = 0.f;
                    }
                for (int v = 0; v < c.n_cat; ++v)
                    if ((v & (P - 1)) == P - 1 - part) *(uint32_t *)(itile + v * 32 * IRS + irow) = 0u;
            }
        }
        __syncthreads();
        const int wave = tid >> 6, lane = tid & 63, m = lane & 31, hf = lane >> 5;
        const bool row_ok = bblk * 32 + m < p.B;
        for (int ll = wave; ll < TL; ll += 4) {
            const int l = lt0 + ll;
            if (l >= L) break;
            for (int kx = 0; kx < p.kx0; ++kx) {
                bf16x8 hi, lo;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int code = chan[16 * kx + 8 * hf + j];
                    float v = 0.f;
                    if (code >= 0x10000) {
                        const int cur = itile[((code >> 8) & 255) * 32 * IRS + m * IRS + ll];
                        v = (!row_ok || (code & 255) == cur) ? 0.f : TD_LOG_1E30;   // mixed_diffusion.py:44-51
                    } else if (code >= 0) {
                        v = ztile[(code * 32 + m) * RS + ll];
                    }
                    const __bf16 h = (__bf16)v;
                    hi[j] = h;
                    lo[j] = (__bf16)(v - (float)h);
                }
                char *o = p.frags + (((size_t)l * p.nbb + bblk) * (2 * p.kx0) + 2 * kx) * TD_FRAG_BYTES + lane * 16;
                *(bf16x8 *)o = hi;
                *(bf16x8 *)(o + TD_FRAG_BYTES) = lo;
            }
        }
    }
}

template <bool TERMS>
static int td_launch_reverse(const TdRevParams &p, hipStream_t stream) {
    const td_diffusion_config *cfg = &p.c;
    bool vec = cfg->seq_len % 4 == 0 && td_al16(p.mo) && td_al16(p.z) && ((uintptr_t)p.idx & 3) == 0 && td_al16(p.noise) &&
               td_al16(p.uniforms) && p.noise_stride % 4 == 0 && p.uni_stride % 4 == 0 && td_al16(p.x0_out) &&
               td_al16(p.mean_out) && td_al16(p.log_theta_out);
    const size_t BL = (size_t)p.B * cfg->seq_len;
    // binary variables (every dataset of the reference, etdiff_train.py:24-29) get their own instantiation: the class
    // loops are unrolled to the bound, and 8 classes x 4 positions of logits cost 100 registers more than 2 x 4
    if constexpr (!TERMS) {
        if (p.frags) {   // the caller asked td_reverse_fusable() first
            // fewer 32 x 32 tiles than CUs: four waves share each item (see above)
            const bool split = (size_t)td_cdiv(cfg->seq_len, 32) * p.nbb < TD_REV_SPLIT_BELOW;
            const bool k2 = td_max_k(cfg) <= 2;
            const int P = split ? 4 : 1;
            const size_t shm = td_rf_shm(cfg, p.kx0, P);
            const dim3 grid(td_cdiv(cfg->seq_len, 32 / P), p.nbb);
            const void *fn = split ? (k2 ? (const void *)td_reverse_step_kernel<4, 2, false, 4> : (const void *)td_reverse_step_kernel<4, TD_KMAX, false, 4>)
                                   : (k2 ? (const void *)td_reverse_step_kernel<4, 2, false, 1> : (const void *)td_reverse_step_kernel<4, TD_KMAX, false, 1>);
            if (shm > 65536) TD_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            void *args[] = {(void *)&p};
            TD_HIP(hipLaunchKernel(fn, grid, dim3(256), args, shm, stream));
            return TD_OK;
        }
    }
    if (td_max_k(cfg) > TD_KMAX)
        hipLaunchKernelGGL((td_reverse_step_kernel<1, TD_KMAX_WIDE, TERMS, 0>), dim3(td_grid(BL)), dim3(256), 0, stream, p);
    else if (vec && td_max_k(cfg) <= 2)
        hipLaunchKernelGGL((td_reverse_step_kernel<4, 2, TERMS, 0>), dim3(td_grid(BL / 4)), dim3(256), 0, stream, p);
    else if (vec)
        hipLaunchKernelGGL((td_reverse_step_kernel<4, TD_KMAX, TERMS, 0>), dim3(td_grid(BL / 4)), dim3(256), 0, stream, p);
    else
        hipLaunchKernelGGL((td_reverse_step_kernel<1, TD_KMAX, TERMS, 0>), dim3(td_grid(BL)), dim3(256), 0, stream, p);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

// Whether the chain's reverse update can also pack the next step's layer-0 fragments (the vectorised kernel's conditions;
// the tile must fit one CU's LDS).
bool td_reverse_fusable(const td_diffusion_config *cfg, const float *mo, const float *z, const uint8_t *idx, const float *noise,
                        const float *uniforms, size_t noise_stride, size_t uni_stride, int kx0) {
    return cfg->seq_len % 4 == 0 && td_al16(mo) && td_al16(z) && ((uintptr_t)idx & 3) == 0 && td_al16(noise) && td_al16(uniforms) &&
           noise_stride % 4 == 0 && uni_stride % 4 == 0 && td_max_k(cfg) <= TD_KMAX && td_rf_shm(cfg, kx0, 1) <= 160 * 1024;
}

int td_launch_reverse_step(const td_diffusion_config *cfg, int batch, const float *mo, float *z, uint8_t *idx, int t_host,
                           const int32_t *t_dev, int32_t *t_next, const float *noise, const float *uniforms, size_t noise_stride,
                           size_t uni_stride, td_rng rng, hipStream_t stream, char *frags, int kx0) {
    TdRevParams p;
    p.frags = frags; p.kx0 = kx0; p.nbb = td_cdiv(batch, 32);
    p.c = *cfg; p.B = batch; p.sumk = td_sum_k(cfg); p.t_host = t_host; p.mo = mo; p.z = z; p.idx = idx; p.t_dev = t_dev;
    p.t_next = t_next;
    p.noise = noise; p.uniforms = uniforms; p.noise_stride = noise_stride; p.uni_stride = uni_stride; p.rng = rng;
    p.x0_out = p.mean_out = p.log_theta_out = nullptr;
    return td_launch_reverse<false>(p, stream);
}

extern "C" int td_reverse_step_mixed(const td_diffusion_config *cfg, int32_t batch, const float *model_out, float *z,
                                     uint8_t *idx, int32_t t_host, const int32_t *t_dev, const float *noise,
                                     const float *uniforms, td_rng rng, void *stream) {
    int rc = td_check_diff(cfg, batch);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(model_out && (z || cfg->n_num == 0) && (idx || cfg->n_cat == 0), "NULL argument");
    TD_REQUIRE(t_dev || (t_host >= 0 && t_host < cfg->timesteps), "t = %d outside [0, %d)", t_host, cfg->timesteps);
    return td_launch_reverse_step(cfg, batch, model_out, z, idx, t_host, t_dev, nullptr, noise, uniforms, 0, 0, rng,
                                  (hipStream_t)stream, nullptr, 0);
}

extern "C" int td_reverse_step_terms(const td_diffusion_config *cfg, int32_t batch, const float *model_out, const float *z,
                                     const uint8_t *idx, int32_t t, float *x0, float *mean, float *log_theta, void *stream) {
    int rc = td_check_diff(cfg, batch);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(model_out && (z || cfg->n_num == 0) && (idx || cfg->n_cat == 0), "NULL argument");
    TD_REQUIRE((cfg->n_num == 0 || (x0 && mean)) && (cfg->n_cat == 0 || log_theta), "NULL output");
    TD_REQUIRE(t >= 0 && t < cfg->timesteps, "t = %d outside [0, %d)", t, cfg->timesteps);
    TdRevParams p;
    p.c = *cfg; p.B = batch; p.sumk = td_sum_k(cfg); p.t_host = t; p.mo = model_out; p.z = (float *)z; p.idx = (uint8_t *)idx;
    p.t_dev = nullptr; p.t_next = nullptr; p.noise = nullptr; p.uniforms = nullptr; p.noise_stride = 0; p.uni_stride = 0;
    p.rng.seed = 0; p.rng.patient0 = 0; p.rng.tag = 0; p.rng.seed_dev = nullptr;
    p.x0_out = x0; p.mean_out = mean; p.log_theta_out = log_theta;
    p.frags = nullptr; p.nbb = 0; p.kx0 = 0;
    return td_launch_reverse<true>(p, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// loss + gradient (mixed_diffusion.py:390-401)
// ---------------------------------------------------------------------------------------------
template <int VEC, int KM>
__global__ __launch_bounds__(256) void td_mixed_loss_kernel(td_diffusion_config c, int B, int sumk, const float *__restrict__ mo,
                                                            const float *__restrict__ noise, const uint8_t *__restrict__ x0_idx,
                                                            const uint8_t *__restrict__ xt_idx, const int64_t *__restrict__ t,
                                                            float lambda, float gscale, float *__restrict__ grad, double *partial) {
    const int L = c.seq_len, C_io = c.n_num + sumk, T = c.timesteps, LQ = L / VEC;
    const size_t total = (size_t)B * LQ;
    const float w_gauss = c.n_num > 0 ? 1.0f / ((float)B * c.n_num * L) : 0.f;
    const float w_multi = c.n_cat > 0 ? lambda / ((float)B * c.n_cat) : 0.f;
    const float lcaT = td_tab(c, TD_TAB_LOG_CUM_ALPHA, T - 1), l1mcaT = td_tab(c, TD_TAB_LOG_1M_CUM_ALPHA, T - 1);
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / LQ), l0 = (int)(i % LQ) * VEC;
        const int tb = (int)t[b];
        float sq = 0.f;
        for (int ch = 0; ch < c.n_num; ++ch) {
            const TdVec<VEC> a = td_ldf<VEC>(mo + ((size_t)b * C_io + ch) * L + l0);
            const TdVec<VEC> n = td_ldf<VEC>(noise + ((size_t)b * c.n_num + ch) * L + l0);
            TdVec<VEC> g;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float d = a.v[e] - n.v[e];
                sq += d * d;                                                               // :364-368
                g.v[e] = gscale * 2.0f * d * w_gauss;
            }
            if (grad) td_stf<VEC>(grad + ((size_t)b * C_io + ch) * L + l0, g);
        }
        const int tm1 = tb > 0 ? tb - 1 : 0;
        const float lca = td_tab(c, TD_TAB_LOG_CUM_ALPHA, tm1), l1mca = td_tab(c, TD_TAB_LOG_1M_CUM_ALPHA, tm1);
        const float la = td_tab(c, TD_TAB_LOG_ALPHA, tb), l1ma = td_tab(c, TD_TAB_LOG_1M_ALPHA, tb);
        const float s_prev = tb == 0 ? 1.0f : td_fexp(lca);
        float term = 0.f, prior = 0.f;
        int off = 0;
        // q(x_t | x_{t-1}) in log space (:255-262) takes two values per class count: the current class or not;
        // so do the true posterior's q(x_{t-1} | x_0) term and the prior KL (x_0 is exactly one-hot): computed once per
        // class count and thread instead of once per class and position
        int lastK = -1;
        float uni = 0.f, w_hit = 0.f, w_miss = 0.f, evt_hit = 0.f, evt_miss = 0.f, prior_hit = 0.f, prior_miss = 0.f, unif_p = 0.f;
        for (int v = 0; v < c.n_cat; ++v) {
            const int K = c.num_classes[v];
            if (K != lastK) {
                lastK = K;
                const float logK = logf((float)K);
                uni = l1mca - logK;
                unif_p = (1.0f - s_prev) / (float)K;
                w_hit = td_flae(0.f + la, l1ma - logK);
                w_miss = td_flae(TD_LOG_1E30 + la, l1ma - logK);
                evt_hit = tb == 0 ? 0.f : td_flae(0.f + lca, uni);
                evt_miss = tb == 0 ? TD_LOG_1E30 : td_flae(TD_LOG_1E30 + lca, uni);
                const float lq_hit = td_flae(0.f + lcaT, l1mcaT - logK), lq_miss = td_flae(TD_LOG_1E30 + lcaT, l1mcaT - logK);
                prior_hit = td_fexp(lq_hit) * (lq_hit + logK);                              // :330-339
                prior_miss = td_fexp(lq_miss) * (lq_miss + logK);
            }
            const float *zp = mo + ((size_t)b * C_io + c.n_num + off) * L + l0;
            int x0[VEC], xt[VEC];
            td_ldb<VEC>(x0_idx + ((size_t)b * c.n_cat + v) * L + l0, x0);
            td_ldb<VEC>(xt_idx + ((size_t)b * c.n_cat + v) * L + l0, xt);
            float lx0h[KM][VEC], um[KM][VEC];
            float m[VEC], se[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) { m[e] = -INFINITY; se[e] = 0.f; }
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
                    const TdVec<VEC> q = td_ldf<VEC>(zp + (size_t)k * L);
#pragma unroll
                    for (int e = 0; e < VEC; ++e) { lx0h[k][e] = q.v[e]; m[e] = fmaxf(m[e], q.v[e]); }
                }
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) se[e] += td_fexp(lx0h[k][e] - m[e]);
                }
            float mt[VEC], mm[VEC], st[VEC], sm[VEC], lset[VEC], lsem[VEC], tgt_sum[VEC], gp[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float lse = m[e] + td_flog(se[e]);
                mt[e] = -INFINITY; mm[e] = -INFINITY; st[e] = 0.f; sm[e] = 0.f; tgt_sum[e] = 0.f; gp[e] = 0.f;
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
                        lx0h[k][e] -= lse;                                                 // :274-281
                        const float w = k == xt[e] ? w_hit : w_miss;
                        const float evm = tb == 0 ? lx0h[k][e] : td_flae(lx0h[k][e] + lca, uni);
                        um[k][e] = evm + w;
                        mt[e] = fmaxf(mt[e], (k == x0[e] ? evt_hit : evt_miss) + w);
                        mm[e] = fmaxf(mm[e], um[k][e]);
                        prior += k == x0[e] ? prior_hit : prior_miss;
                    }
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
                        const float utk = (k == x0[e] ? evt_hit : evt_miss) + (k == xt[e] ? w_hit : w_miss);
                        st[e] += td_fexp(utk - mt[e]);
                        sm[e] += td_fexp(um[k][e] - mm[e]);
                    }
                lset[e] = mt[e] + td_flog(st[e]);     // direct per-slice logsumexp (vs :32-42)
                lsem[e] = mm[e] + td_flog(sm[e]);
            }
            // um[k] <- theta_k; target probability and the loss term
            float tg[KM][VEC];
#pragma unroll
            for (int k = 0; k < KM; ++k)
                if (k < K) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const float utk = (k == x0[e] ? evt_hit : evt_miss) + (k == xt[e] ? w_hit : w_miss);
                        const float lt = utk - lset[e], lm = um[k][e] - lsem[e];
                        const float p0 = k == x0[e] ? 1.0f : 1e-30f;                        // exp(log x_0), :44-51
                        const float pt = td_fexp(lt);
                        if (tb == 0) term -= p0 * lm;                                       // :351-352
                        else term += pt * (lt - lm);                                        // :348-349
                        tg[k][e] = tb == 0 ? p0 : pt;
                        um[k][e] = td_fexp(lm);
                        tgt_sum[e] += tg[k][e];
                    }
                }
            if (grad) {
                const float scale = gscale * w_multi * (float)T;
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const float pk = td_fexp(lx0h[k][e]);
                            const float pi = tb == 0 ? pk : s_prev * pk + unif_p;
                            const float gk = s_prev * (um[k][e] * tgt_sum[e] - tg[k][e]) * __builtin_amdgcn_rcpf(pi);
                            lx0h[k][e] = pk;
                            tg[k][e] = gk;
                            gp[e] += gk * pk;
                        }
                    }
#pragma unroll
                for (int k = 0; k < KM; ++k)
                    if (k < K) {
                        TdVec<VEC> o;
#pragma unroll
                        for (int e = 0; e < VEC; ++e) o.v[e] = scale * lx0h[k][e] * (tg[k][e] - gp[e]);
                        td_stf<VEC>(grad + ((size_t)b * C_io + c.n_num + off + k) * L + l0, o);
                    }
            }
            off += K;
        }
        local += (double)(sq * w_gauss) + (double)w_multi * ((double)term * T + (double)prior);
    }
    // block reduction (deterministic order)
    __shared__ double red[256];
    red[threadIdx.x] = local;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void td_loss_finish_kernel(const double *partial, int n, float scale, float *loss) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) s += partial[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int sft = 128; sft > 0; sft >>= 1) {
        if ((int)threadIdx.x < sft) red[threadIdx.x] += red[threadIdx.x + sft];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = (float)(red[0] * scale);
}

#define TD_LOSS_BLOCKS 1024

extern "C" size_t td_mixed_loss_workspace_bytes(const td_diffusion_config *, int32_t) {
    return TD_LOSS_BLOCKS * sizeof(double);
}

// both loss entry points: the fused loss + gradient pass into `partial`, then the finishing reduction
static int td_mixed_loss_launch(const td_diffusion_config *cfg, int batch, const float *model_out, const float *noise,
                                const uint8_t *x0_idx, const uint8_t *xt_idx, const int64_t *t, float loss_lambda, float scale,
                                float *loss, float *grad, double *partial, hipStream_t stream) {
    const bool vec = cfg->seq_len % 4 == 0 && td_al16(model_out) && td_al16(noise) && td_al16(grad) && ((uintptr_t)x0_idx & 3) == 0 &&
                     ((uintptr_t)xt_idx & 3) == 0;
    const size_t BL = (size_t)batch * cfg->seq_len;
    const int sumk = td_sum_k(cfg);
    int blocks;
    if (td_max_k(cfg) > TD_KMAX) {
        blocks = td_grid(BL, TD_LOSS_BLOCKS);
        hipLaunchKernelGGL((td_mixed_loss_kernel<1, TD_KMAX_WIDE>), dim3(blocks), dim3(256), 0, stream, *cfg, batch, sumk, model_out,
                           noise, x0_idx, xt_idx, t, loss_lambda, scale, grad, partial);
    } else if (vec && td_max_k(cfg) <= 2) {
        blocks = td_grid(BL / 4, TD_LOSS_BLOCKS);
        hipLaunchKernelGGL((td_mixed_loss_kernel<4, 2>), dim3(blocks), dim3(256), 0, stream, *cfg, batch, sumk, model_out, noise,
                           x0_idx, xt_idx, t, loss_lambda, scale, grad, partial);
    } else if (vec) {
        blocks = td_grid(BL / 4, TD_LOSS_BLOCKS);
        hipLaunchKernelGGL((td_mixed_loss_kernel<4, TD_KMAX>), dim3(blocks), dim3(256), 0, stream, *cfg, batch, sumk, model_out, noise,
                           x0_idx, xt_idx, t, loss_lambda, scale, grad, partial);
    } else {
        blocks = td_grid(BL, TD_LOSS_BLOCKS);
        hipLaunchKernelGGL((td_mixed_loss_kernel<1, TD_KMAX>), dim3(blocks), dim3(256), 0, stream, *cfg, batch, sumk, model_out, noise,
                           x0_idx, xt_idx, t, loss_lambda, scale, grad, partial);
    }
    TD_LAUNCH_CHECK();
    hipLaunchKernelGGL(td_loss_finish_kernel, dim3(1), dim3(256), 0, stream, (const double *)partial, blocks, 1.0f, loss);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

extern "C" int td_mixed_loss(const td_diffusion_config *cfg, int32_t batch, const float *model_out, const float *noise,
                             const uint8_t *x0_idx, const uint8_t *xt_idx, const int64_t *t, float loss_lambda,
                             float scale, float *loss, float *grad, void *workspace, void *stream) {
    int rc = td_check_diff(cfg, batch);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(model_out && t && loss && workspace, "NULL argument");
    TD_REQUIRE(cfg->n_num == 0 || noise, "noise required when n_num > 0");
    TD_REQUIRE(cfg->n_cat == 0 || (x0_idx && xt_idx), "x0_idx / xt_idx required when n_cat > 0");
    return td_mixed_loss_launch(cfg, batch, model_out, noise, x0_idx, xt_idx, t, loss_lambda, scale, loss, grad,
                                (double *)workspace, (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// td_mixed_loss for callers that cannot allocate (HIP graph capture, tight serving loops, harnesses that forbid
// hipMalloc): the block partials live in a file-scope device array instead of a caller workspace.  Each GPU the
// library runs on holds its own copy of the array (8 KiB) until the process ends; its device address is looked up
// once per device with that device current (the array's NAME is a host-side stand-in, never a kernel argument).
// Every call on a GPU shares that GPU's array: calls that overlap on one GPU (different streams) race.  Nothing in
// it has to start at zero: each launch overwrites the `blocks` partials the finishing kernel then reads.
// ---------------------------------------------------------------------------------------------
static __device__ __align__(256) double td_loss_scratch[TD_LOSS_BLOCKS];
#define TD_NOALLOC_MAX_DEVICES 64
static double *g_loss_scratch_ptr[TD_NOALLOC_MAX_DEVICES];   // per device: the address of ITS copy of td_loss_scratch

extern "C" int td_mixed_loss_noalloc(const td_diffusion_config *cfg, int32_t batch, const float *model_out,
                                     const float *noise, const uint8_t *x0_idx, const uint8_t *xt_idx, const int64_t *t,
                                     float loss_lambda, float scale, float *loss, float *grad, void *stream) {
    int rc = td_check_diff(cfg, batch);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(model_out && t && loss, "NULL argument");
    TD_REQUIRE(cfg->n_num == 0 || noise, "noise required when n_num > 0");
    TD_REQUIRE(cfg->n_cat == 0 || (x0_idx && xt_idx), "x0_idx / xt_idx required when n_cat > 0");
    const size_t positions = (size_t)batch * cfg->seq_len;
    if (positions > (size_t)TD_NOALLOC_MAX_POSITIONS) {
        td_set_error("td_mixed_loss_noalloc: batch * seq_len = %zu exceeds the maximum of %zu positions; use td_mixed_loss",
                     positions, (size_t)TD_NOALLOC_MAX_POSITIONS);
        return TD_ERR_UNSUPPORTED;
    }
    int dev = -1;
    TD_HIP(hipGetDevice(&dev));
    TD_REQUIRE(dev >= 0 && dev < TD_NOALLOC_MAX_DEVICES, "device ordinal %d out of range", dev);
    if (!g_loss_scratch_ptr[dev]) {   // first call on this device (not under graph capture: call once beforehand)
        double *ptr = nullptr;
        TD_HIP(hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(td_loss_scratch)));
        TD_REQUIRE(ptr != nullptr, "hipGetSymbolAddress returned NULL for the loss scratch");
        g_loss_scratch_ptr[dev] = ptr;
    }
    return td_mixed_loss_launch(cfg, batch, model_out, noise, x0_idx, xt_idx, t, loss_lambda, scale, loss, grad,
                                g_loss_scratch_ptr[dev], (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// sampler pieces (mixed_diffusion.py:425-454)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void td_prior_kernel(td_diffusion_config c, int B, int sumk, const float *noise,
                                                       const float *uniforms, td_rng rng, float *z, uint8_t *idx) {
    const int L = c.seq_len;
    const size_t total = (size_t)B * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / L), l = (int)(i % L);
        TdDraw<1> draw(noise, uniforms, rng, b, l, c.n_num, sumk, L, (uint32_t)c.timesteps);
        for (int ch = 0; ch < c.n_num; ++ch) z[((size_t)b * c.n_num + ch) * L + l] = draw.normal(ch).v[0];   // :429
        int off = 0;
        for (int v = 0; v < c.n_cat; ++v) {   // :434-435: Gumbel-max over all-zero logits
            int best = 0;
            float bv = -INFINITY;
            for (int k = 0; k < c.num_classes[v]; ++k) {
                const float val = td_gumbel(draw.uniform(off + k).v[0]) + 0.0f;
                if (val > bv) { bv = val; best = k; }
            }
            idx[((size_t)b * c.n_cat + v) * L + l] = (uint8_t)best;
            off += c.num_classes[v];
        }
    }
}

extern "C" int td_sample_prior(const td_diffusion_config *cfg, int32_t batch, const float *noise, const float *uniforms,
                               td_rng rng, float *z, uint8_t *idx, void *stream) {
    int rc = td_check_diff(cfg, batch);
    if (rc != TD_OK) return rc;
    TD_REQUIRE((z || cfg->n_num == 0) && (idx || cfg->n_cat == 0), "NULL argument");
    hipLaunchKernelGGL(td_prior_kernel, dim3(td_grid((size_t)batch * cfg->seq_len)), dim3(256), 0, (hipStream_t)stream, *cfg,
                       batch, td_sum_k(cfg), noise, uniforms, rng, z, idx);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

__global__ __launch_bounds__(256) void td_build_input_kernel(td_diffusion_config c, int B, int sumk, const float *z,
                                                             const uint8_t *idx, float *x_in) {
    const int L = c.seq_len, C_io = c.n_num + sumk;
    const size_t total = (size_t)B * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / L), l = (int)(i % L);
        for (int ch = 0; ch < c.n_num; ++ch) x_in[((size_t)b * C_io + ch) * L + l] = z[((size_t)b * c.n_num + ch) * L + l];
        int off = 0;
        for (int v = 0; v < c.n_cat; ++v) {
            const int cur = idx[((size_t)b * c.n_cat + v) * L + l];
            for (int k = 0; k < c.num_classes[v]; ++k)
                x_in[((size_t)b * C_io + c.n_num + off + k) * L + l] = k == cur ? 0.f : TD_LOG_1E30;   // :44-51
            off += c.num_classes[v];
        }
    }
}

extern "C" int td_build_denoiser_input(const td_diffusion_config *cfg, int32_t batch, const float *z, const uint8_t *idx,
                                       float *x_in, void *stream) {
    int rc = td_check_diff(cfg, batch);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(x_in && (z || cfg->n_num == 0) && (idx || cfg->n_cat == 0), "NULL argument");
    hipLaunchKernelGGL(td_build_input_kernel, dim3(td_grid((size_t)batch * cfg->seq_len)), dim3(256), 0, (hipStream_t)stream,
                       *cfg, batch, td_sum_k(cfg), z, idx, x_in);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

__global__ __launch_bounds__(256) void td_decode_kernel(td_diffusion_config c, int B, const float *z, const uint8_t *idx,
                                                        float *samples) {
    const int L = c.seq_len;
    const size_t total = (size_t)B * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / L), l = (int)(i % L);
        for (int ch = 0; ch < c.n_num; ++ch)
        {
            const float zv = z[((size_t)b * c.n_num + ch) * L + l];
            samples[((size_t)b * c.data_channels + c.num_index[ch]) * L + l] = c.raw_range ? zv : (zv + 1.0f) * 0.5f;   // :449
        }
        for (int v = 0; v < c.n_cat; ++v)
            samples[((size_t)b * c.data_channels + c.cat_index[v]) * L + l] = (float)idx[((size_t)b * c.n_cat + v) * L + l];
    }
}

extern "C" int td_decode_samples(const td_diffusion_config *cfg, int32_t batch, const float *z, const uint8_t *idx,
                                 float *samples, void *stream) {
    int rc = td_check_diff(cfg, batch);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(samples && (z || cfg->n_num == 0) && (idx || cfg->n_cat == 0), "NULL argument");
    // channels the index lists do not cover stay zero (create_samples, :173-179)
    TD_HIP(hipMemsetAsync(samples, 0, (size_t)batch * cfg->data_channels * cfg->seq_len * sizeof(float), (hipStream_t)stream));
    hipLaunchKernelGGL(td_decode_kernel, dim3(td_grid((size_t)batch * cfg->seq_len)), dim3(256), 0, (hipStream_t)stream, *cfg,
                       batch, z, idx, samples);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

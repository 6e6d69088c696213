// Mixed-diffusion arithmetic as fused elementwise passes (one thread per (sequence, position)).
// Reference: models/ETDiff/mixed_diffusion.py; n_cat = 0 covers models/ETDiff/gaussian_diffusion.py.
// All of these kernels are HBM-bound: consecutive lanes walk consecutive positions l of one channel
// row, so every load/store instruction is a coalesced 256-byte row segment.
#include "common.h"

#define TD_KMAX 8  // classes per discrete variable held in registers

struct TdTabs {
    const float *t[TD_TAB_COUNT];
};

static int td_check_diff(const td_diffusion_config *c, int batch) {
    TD_REQUIRE(c != nullptr, "td_diffusion_config is NULL");
    TD_REQUIRE(batch > 0, "batch must be positive");
    TD_REQUIRE(c->n_num >= 0 && c->n_num <= 2 * TD_MAX_CAT, "n_num = %d out of range", c->n_num);
    TD_REQUIRE(c->n_cat >= 0 && c->n_cat <= TD_MAX_CAT, "n_cat = %d out of range", c->n_cat);
    TD_REQUIRE(c->n_num + c->n_cat > 0, "no channels");
    TD_REQUIRE(c->seq_len > 0 && c->timesteps > 0, "seq_len and timesteps must be positive");
    TD_REQUIRE(c->tables != nullptr, "tables is NULL");
    for (int v = 0; v < c->n_cat; ++v)
        if (c->num_classes[v] < 1 || c->num_classes[v] > TD_KMAX) {
            td_set_error("num_classes[%d] = %d not in [1, %d]", v, c->num_classes[v], TD_KMAX);
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

__device__ __forceinline__ float td_tab(const td_diffusion_config &c, int row, int t) {
    return c.tables[(size_t)row * c.timesteps + t];
}

__device__ __forceinline__ float td_gumbel(float u) { return -logf(-logf(u + 1e-30f) + 1e-30f); }

// noise source for one (b, l): external arrays or Philox
struct TdDraw {
    const float *noise, *uniforms;
    uint64_t seed;
    uint32_t patient, l, tag;
    size_t nbase, ubase;  // offsets of (b, channel 0, l)
    int L;
    float nbuf[4], ubuf[4];
    int nblk, ublk;
    __device__ TdDraw(const float *n, const float *u, td_rng rng, int b, int l_, int n_num, int sumk, int L_, uint32_t tag_)
        : noise(n), uniforms(u), seed(rng.seed), patient(rng.patient0 + (uint32_t)b), l((uint32_t)l_), tag(tag_),
          nbase(((size_t)b * n_num) * L_ + l_), ubase(((size_t)b * sumk) * L_ + l_), L(L_), nblk(-1), ublk(-1) {}
    __device__ float normal(int ch) {
        if (noise) return noise[nbase + (size_t)ch * L];
        if ((ch >> 2) != nblk) { nblk = ch >> 2; td_normal4(seed, patient, l, tag, (uint32_t)nblk, nbuf); }
        const int e = ch & 3;
        return e == 0 ? nbuf[0] : e == 1 ? nbuf[1] : e == 2 ? nbuf[2] : nbuf[3];
    }
    __device__ float uniform(int slot) {
        if (uniforms) return uniforms[ubase + (size_t)slot * L];
        if ((slot >> 2) != ublk) { ublk = slot >> 2; td_uniform4(seed, patient, l, tag, (uint32_t)ublk, ubuf); }
        const int e = slot & 3;
        return e == 0 ? ubuf[0] : e == 1 ? ubuf[1] : e == 2 ? ubuf[2] : ubuf[3];
    }
};

// ---------------------------------------------------------------------------------------------
// forward noising for a training batch (mixed_diffusion.py:403-409, :375-387)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void td_q_sample_kernel(td_diffusion_config c, int B, int sumk, const float *data,
                                                          const int64_t *t, const float *noise, const float *uniforms,
                                                          td_rng rng, float *x_in, uint8_t *x0_idx, uint8_t *xt_idx,
                                                          float *noise_out) {
    const int L = c.seq_len, C_io = c.n_num + sumk;
    const size_t total = (size_t)B * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / L), l = (int)(i % L);
        const int tb = (int)t[b];
        TdDraw draw(noise, uniforms, rng, b, l, c.n_num, sumk, L, rng.tag);
        const float sa = td_tab(c, TD_TAB_SQRT_AC, tb), s1 = td_tab(c, TD_TAB_SQRT_1M_AC, tb);
        for (int ch = 0; ch < c.n_num; ++ch) {
            const float x0 = data[((size_t)b * c.data_channels + c.num_index[ch]) * L + l] * 2.0f - 1.0f;  // :189-194
            const float eps = draw.normal(ch);
            x_in[((size_t)b * C_io + ch) * L + l] = sa * x0 + s1 * eps;  // :240-248
            if (noise_out) noise_out[((size_t)b * c.n_num + ch) * L + l] = eps;
        }
        const float lca = td_tab(c, TD_TAB_LOG_CUM_ALPHA, tb), l1mca = td_tab(c, TD_TAB_LOG_1M_CUM_ALPHA, tb);
        int off = 0;
        for (int v = 0; v < c.n_cat; ++v) {
            const int K = c.num_classes[v];
            const float logK = logf((float)K);
            const int x0 = (int)data[((size_t)b * c.data_channels + c.cat_index[v]) * L + l];
            int best = 0;
            float bv = -INFINITY;
            for (int k = 0; k < K; ++k) {
                const float lx = k == x0 ? 0.f : TD_LOG_1E30;
                const float logit = td_log_add_exp(lx + lca, l1mca - logK);  // :264-272
                const float val = td_gumbel(draw.uniform(off + k)) + logit;  // :322-324
                if (val > bv) { bv = val; best = k; }
            }
            x0_idx[((size_t)b * c.n_cat + v) * L + l] = (uint8_t)x0;
            xt_idx[((size_t)b * c.n_cat + v) * L + l] = (uint8_t)best;
            for (int k = 0; k < K; ++k) x_in[((size_t)b * C_io + c.n_num + off + k) * L + l] = k == best ? 0.f : TD_LOG_1E30;
            off += K;
        }
    }
}

static inline int td_grid(size_t total, int cap = 4096) {
    size_t b = (total + 255) / 256;
    return (int)(b > (size_t)cap ? cap : (b ? b : 1));
}

extern "C" int td_q_sample_mixed(const td_diffusion_config *cfg, int32_t batch, const float *data, const int64_t *t,
                                 const float *noise, const float *uniforms, td_rng rng, float *x_in, uint8_t *x0_idx,
                                 uint8_t *xt_idx, float *noise_out, void *stream) {
    int rc = td_check_diff(cfg, batch);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(data && t && x_in, "NULL argument");
    TD_REQUIRE(cfg->n_cat == 0 || (x0_idx && xt_idx), "x0_idx / xt_idx required when n_cat > 0");
    hipLaunchKernelGGL(td_q_sample_kernel, dim3(td_grid((size_t)batch * cfg->seq_len)), dim3(256), 0, (hipStream_t)stream,
                       *cfg, batch, td_sum_k(cfg), data, t, noise, uniforms, rng, x_in, x0_idx, xt_idx, noise_out);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

// ---------------------------------------------------------------------------------------------
// reverse update (mixed_diffusion.py:412-423, :307-311)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void td_reverse_step_kernel(td_diffusion_config c, int B, int sumk, const float *mo,
                                                              float *z, uint8_t *idx, int t_host, const int32_t *t_dev,
                                                              const float *noise, const float *uniforms,
                                                              size_t noise_stride, size_t uni_stride, td_rng rng) {
    const int L = c.seq_len, C_io = c.n_num + sumk, T = c.timesteps;
    const int t = t_dev ? t_dev[0] : t_host;
    if (t < 0 || t >= T) return;
    // chained external draws: slab 0 is the prior, slab T - t belongs to step t
    if (noise && noise_stride) noise += (size_t)(T - t) * noise_stride;
    if (uniforms && uni_stride) uniforms += (size_t)(T - t) * uni_stride;
    const float c_recip = td_tab(c, TD_TAB_SQRT_RECIP_AC, t), c_recipm1 = td_tab(c, TD_TAB_SQRT_RECIPM1_AC, t);
    const float c1 = td_tab(c, TD_TAB_POST_C1, t), c2 = td_tab(c, TD_TAB_POST_C2, t);
    const float sigma = expf(0.5f * td_tab(c, TD_TAB_POST_LOGVAR, t));
    const int tm1 = t > 0 ? t - 1 : 0;
    const float lca = td_tab(c, TD_TAB_LOG_CUM_ALPHA, tm1), l1mca = td_tab(c, TD_TAB_LOG_1M_CUM_ALPHA, tm1);
    const float la = td_tab(c, TD_TAB_LOG_ALPHA, t), l1ma = td_tab(c, TD_TAB_LOG_1M_ALPHA, t);
    const size_t total = (size_t)B * L;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / L), l = (int)(i % L);
        TdDraw draw(noise, uniforms, rng, b, l, c.n_num, sumk, L, (uint32_t)t);
        for (int ch = 0; ch < c.n_num; ++ch) {
            const size_t zi = ((size_t)b * c.n_num + ch) * L + l;
            const float eps = mo[((size_t)b * C_io + ch) * L + l], zt = z[zi];
            float x0 = c_recip * zt - c_recipm1 * eps;          // :196-201
            x0 = fminf(fmaxf(x0, -1.0f), 1.0f);                  // :234-235
            const float mean = c1 * x0 + c2 * zt;                // :210-218
            const float nz = t > 0 ? draw.normal(ch) : 0.f;      // :421
            z[zi] = mean + sigma * nz;                            // :422
        }
        int off = 0;
        // q(x_t | x_{t-1}) in log space (:255-262) takes two values per class count: the current class or not
        int lastK = -1;
        float logK = 0.f, w_hit = 0.f, w_miss = 0.f;
        for (int v = 0; v < c.n_cat; ++v) {
            const int K = c.num_classes[v];
            if (K != lastK) {
                lastK = K;
                logK = logf((float)K);
                w_hit = td_log_add_exp(0.f + la, l1ma - logK);
                w_miss = td_log_add_exp(TD_LOG_1E30 + la, l1ma - logK);
            }
            const float *zp = mo + ((size_t)b * C_io + c.n_num + off) * L + l;
            float logit[TD_KMAX];
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < TD_KMAX; ++k) {
                logit[k] = k < K ? zp[(size_t)k * L] : -INFINITY;
                m = fmaxf(m, logit[k]);
            }
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < TD_KMAX; ++k) se += k < K ? expf(logit[k] - m) : 0.f;
            const float lse = m + logf(se);                      // :274-281
            const int xt = idx[((size_t)b * c.n_cat + v) * L + l];
            int best = 0;
            float bv = -INFINITY;
#pragma unroll
            for (int k = 0; k < TD_KMAX; ++k) {
                if (k < K) {
                    const float lx0 = logit[k] - lse;
                    const float ev = t == 0 ? lx0 : td_log_add_exp(lx0 + lca, l1mca - logK);   // :284-290
                    const float w = k == xt ? w_hit : w_miss;
                    // the per-slice normaliser (:293) shifts all K values equally: argmax-invariant
                    const float val = td_gumbel(draw.uniform(off + k)) + (ev + w);
                    if (val > bv) { bv = val; best = k; }
                }
            }
            idx[((size_t)b * c.n_cat + v) * L + l] = (uint8_t)best;
            off += K;
        }
    }
}

int td_launch_reverse_step(const td_diffusion_config *cfg, int batch, const float *mo, float *z, uint8_t *idx, int t_host,
                           const int32_t *t_dev, const float *noise, const float *uniforms, size_t noise_stride,
                           size_t uni_stride, td_rng rng, hipStream_t stream) {
    hipLaunchKernelGGL(td_reverse_step_kernel, dim3(td_grid((size_t)batch * cfg->seq_len)), dim3(256), 0, stream, *cfg,
                       batch, td_sum_k(cfg), mo, z, idx, t_host, t_dev, noise, uniforms, noise_stride, uni_stride, rng);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

extern "C" int td_reverse_step_mixed(const td_diffusion_config *cfg, int32_t batch, const float *model_out, float *z,
                                     uint8_t *idx, int32_t t_host, const int32_t *t_dev, const float *noise,
                                     const float *uniforms, td_rng rng, void *stream) {
    int rc = td_check_diff(cfg, batch);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(model_out && (z || cfg->n_num == 0) && (idx || cfg->n_cat == 0), "NULL argument");
    TD_REQUIRE(t_dev || (t_host >= 0 && t_host < cfg->timesteps), "t = %d outside [0, %d)", t_host, cfg->timesteps);
    return td_launch_reverse_step(cfg, batch, model_out, z, idx, t_host, t_dev, noise, uniforms, 0, 0, rng,
                                  (hipStream_t)stream);
}

// ---------------------------------------------------------------------------------------------
// loss + gradient (mixed_diffusion.py:390-401)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void td_mixed_loss_kernel(td_diffusion_config c, int B, int sumk, const float *mo,
                                                            const float *noise, const uint8_t *x0_idx,
                                                            const uint8_t *xt_idx, const int64_t *t, float lambda,
                                                            float gscale, float *grad, double *partial) {
    const int L = c.seq_len, C_io = c.n_num + sumk, T = c.timesteps;
    const size_t total = (size_t)B * L;
    const float w_gauss = c.n_num > 0 ? 1.0f / ((float)B * c.n_num * L) : 0.f;
    const float w_multi = c.n_cat > 0 ? lambda / ((float)B * c.n_cat) : 0.f;
    const float lcaT = td_tab(c, TD_TAB_LOG_CUM_ALPHA, T - 1), l1mcaT = td_tab(c, TD_TAB_LOG_1M_CUM_ALPHA, T - 1);
    double local = 0.0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int b = (int)(i / L), l = (int)(i % L);
        const int tb = (int)t[b];
        float sq = 0.f;
        for (int ch = 0; ch < c.n_num; ++ch) {
            const float d = mo[((size_t)b * C_io + ch) * L + l] - noise[((size_t)b * c.n_num + ch) * L + l];
            sq += d * d;                                                                   // :364-368
            if (grad) grad[((size_t)b * C_io + ch) * L + l] = gscale * 2.0f * d * w_gauss;
        }
        const int tm1 = tb > 0 ? tb - 1 : 0;
        const float lca = td_tab(c, TD_TAB_LOG_CUM_ALPHA, tm1), l1mca = td_tab(c, TD_TAB_LOG_1M_CUM_ALPHA, tm1);
        const float la = td_tab(c, TD_TAB_LOG_ALPHA, tb), l1ma = td_tab(c, TD_TAB_LOG_1M_ALPHA, tb);
        const float s_prev = tb == 0 ? 1.0f : expf(lca);
        float term = 0.f, prior = 0.f;
        int off = 0;
        // q(x_t | x_{t-1}) in log space (:255-262) takes two values per class count: the current class or not
        // ... and so do the true posterior's q(x_{t-1} | x_0) term and the prior KL (x_0 is exactly one-hot): computed
        // once per class count instead of once per class -- the same expressions on the same inputs, bit for bit
        int lastK = -1;
        float logK = 0.f, w_hit = 0.f, w_miss = 0.f, evt_hit = 0.f, evt_miss = 0.f, prior_hit = 0.f, prior_miss = 0.f;
        for (int v = 0; v < c.n_cat; ++v) {
            const int K = c.num_classes[v];
            if (K != lastK) {
                lastK = K;
                logK = logf((float)K);
                w_hit = td_log_add_exp(0.f + la, l1ma - logK);
                w_miss = td_log_add_exp(TD_LOG_1E30 + la, l1ma - logK);
                evt_hit = tb == 0 ? 0.f : td_log_add_exp(0.f + lca, l1mca - logK);
                evt_miss = tb == 0 ? TD_LOG_1E30 : td_log_add_exp(TD_LOG_1E30 + lca, l1mca - logK);
                const float lq_hit = td_log_add_exp(0.f + lcaT, l1mcaT - logK), lq_miss = td_log_add_exp(TD_LOG_1E30 + lcaT, l1mcaT - logK);
                prior_hit = expf(lq_hit) * (lq_hit + logK);                                // :330-339
                prior_miss = expf(lq_miss) * (lq_miss + logK);
            }
            const float *zp = mo + ((size_t)b * C_io + c.n_num + off) * L + l;
            const int x0 = x0_idx[((size_t)b * c.n_cat + v) * L + l], xt = xt_idx[((size_t)b * c.n_cat + v) * L + l];
            float lx0h[TD_KMAX], ut[TD_KMAX], um[TD_KMAX];
            float m = -INFINITY;
#pragma unroll
            for (int k = 0; k < TD_KMAX; ++k) {
                lx0h[k] = k < K ? zp[(size_t)k * L] : -INFINITY;
                m = fmaxf(m, lx0h[k]);
            }
            float se = 0.f;
#pragma unroll
            for (int k = 0; k < TD_KMAX; ++k)
                if (k < K) se += expf(lx0h[k] - m);
            const float lse = m + logf(se);
            float mt = -INFINITY, mm = -INFINITY;
#pragma unroll
            for (int k = 0; k < TD_KMAX; ++k) {
                if (k < K) {
                    lx0h[k] -= lse;                                                        // :274-281
                    const float w = k == xt ? w_hit : w_miss;
                    const float evt = k == x0 ? evt_hit : evt_miss;
                    const float evm = tb == 0 ? lx0h[k] : td_log_add_exp(lx0h[k] + lca, l1mca - logK);
                    ut[k] = evt + w;
                    um[k] = evm + w;
                    mt = fmaxf(mt, ut[k]);
                    mm = fmaxf(mm, um[k]);
                    prior += k == x0 ? prior_hit : prior_miss;
                } else {
                    ut[k] = -INFINITY; um[k] = -INFINITY;
                }
            }
            float st = 0.f, sm = 0.f;
#pragma unroll
            for (int k = 0; k < TD_KMAX; ++k)
                if (k < K) {
                    st += expf(ut[k] - mt);
                    sm += expf(um[k] - mm);
                }
            const float lset = mt + logf(st), lsem = mm + logf(sm);  // direct per-slice logsumexp (vs :32-42)
            float tgt_sum = 0.f, gp = 0.f;
            float gk[TD_KMAX];
#pragma unroll
            for (int k = 0; k < TD_KMAX; ++k) {
                if (k < K) {
                    const float lt = ut[k] - lset, lm = um[k] - lsem;
                    const float lx0 = k == x0 ? 0.f : TD_LOG_1E30;
                    if (tb == 0) term -= expf(lx0) * lm;                                   // :351-352
                    else term += expf(lt) * (lt - lm);                                     // :348-349
                    ut[k] = tb == 0 ? expf(lx0) : expf(lt);   // target probability
                    um[k] = expf(lm);                          // theta
                    tgt_sum += ut[k];
                }
            }
            if (grad) {
                const float scale = gscale * w_multi * (float)T;
#pragma unroll
                for (int k = 0; k < TD_KMAX; ++k) {
                    if (k < K) {
                        const float pk = expf(lx0h[k]);
                        const float pi = tb == 0 ? pk : s_prev * pk + (1.0f - s_prev) / (float)K;
                        gk[k] = s_prev * (um[k] * tgt_sum - ut[k]) / pi;
                        gp += gk[k] * pk;
                    }
                }
#pragma unroll
                for (int k = 0; k < TD_KMAX; ++k)
                    if (k < K) grad[((size_t)b * C_io + c.n_num + off + k) * L + l] = scale * expf(lx0h[k]) * (gk[k] - gp);
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

extern "C" int td_mixed_loss(const td_diffusion_config *cfg, int32_t batch, const float *model_out, const float *noise,
                             const uint8_t *x0_idx, const uint8_t *xt_idx, const int64_t *t, float loss_lambda,
                             float scale, float *loss, float *grad, void *workspace, void *stream) {
    int rc = td_check_diff(cfg, batch);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(model_out && t && loss && workspace, "NULL argument");
    TD_REQUIRE(cfg->n_num == 0 || noise, "noise required when n_num > 0");
    TD_REQUIRE(cfg->n_cat == 0 || (x0_idx && xt_idx), "x0_idx / xt_idx required when n_cat > 0");
    const int blocks = td_grid((size_t)batch * cfg->seq_len, TD_LOSS_BLOCKS);
    hipLaunchKernelGGL(td_mixed_loss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *cfg, batch, td_sum_k(cfg),
                       model_out, noise, x0_idx, xt_idx, t, loss_lambda, scale, grad, (double *)workspace);
    TD_LAUNCH_CHECK();
    hipLaunchKernelGGL(td_loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)workspace, blocks,
                       1.0f, loss);
    TD_LAUNCH_CHECK();
    return TD_OK;
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
    double *partial = g_loss_scratch_ptr[dev];
    const int blocks = td_grid(positions, TD_LOSS_BLOCKS);
    hipLaunchKernelGGL(td_mixed_loss_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, *cfg, batch, td_sum_k(cfg),
                       model_out, noise, x0_idx, xt_idx, t, loss_lambda, scale, grad, partial);
    TD_LAUNCH_CHECK();
    hipLaunchKernelGGL(td_loss_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const double *)partial, blocks,
                       1.0f, loss);
    TD_LAUNCH_CHECK();
    return TD_OK;
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
        TdDraw draw(noise, uniforms, rng, b, l, c.n_num, sumk, L, (uint32_t)c.timesteps);
        for (int ch = 0; ch < c.n_num; ++ch) z[((size_t)b * c.n_num + ch) * L + l] = draw.normal(ch);   // :429
        int off = 0;
        for (int v = 0; v < c.n_cat; ++v) {   // :434-435: Gumbel-max over all-zero logits
            int best = 0;
            float bv = -INFINITY;
            for (int k = 0; k < c.num_classes[v]; ++k) {
                const float val = td_gumbel(draw.uniform(off + k)) + 0.0f;
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
            samples[((size_t)b * c.data_channels + c.num_index[ch]) * L + l] = (z[((size_t)b * c.n_num + ch) * L + l] + 1.0f) * 0.5f;
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

// Denoiser forward: weight packing, time embedding + FiLM, layer-0 input packing, the recurrent
// layers (rnn_layer.h) and the LayerNorm + FiLM + fc head.  Reference: models/ETDiff/blocks.py.
#include "rnn_launch.h"
#include "rnn_geom.h"
#include "rnn_cluster.h"

#include <math.h>

int td_layer_kernel_mode(void);   // api.cpp
int td_head_fusion_mode(void);    // api.cpp
#ifndef TD_CLUSTER_AUTO_BLOCKS
#define TD_CLUSTER_AUTO_BLOCKS 128   // x 32 sequences, both directions counted: 2048 bidirectional sequences (training step there 49.4 ms against 52.6 ms with the unit-split forward; equal at 4096)
#endif
#ifndef TD_RESIDENT_AUTO_WORKGROUPS
#define TD_RESIDENT_AUTO_WORKGROUPS 1024   // H <= 96, one workgroup per (batch block, direction): up to this many of them (16384 sequences bidirectional: there the sampler step equals the batch-block family's with its fused head, 0.80 against 0.82 ms at H = 92; at 131072 it loses, 6.78 against 6.25 ms)
#endif
// rnn_cluster.hip
size_t td_cluster_xch_dir_bytes(const TdRnnGeom &g, int nbb);
bool td_cluster_supported(const TdRnnGeom &g, int nbb, int L);
int td_cluster_pack_all(const td_rnn_config *cfg, const TdRnnGeom &g, const float *const *lw, char *packed, hipStream_t stream);
int td_launch_layer_cluster(int gru, int save, const TdRnnGeom &g, const TdLayerParams &lp, const char *wcluster, size_t dir_stride,
                            bool layer0, unsigned long long *xch, hipStream_t stream);

extern "C" size_t td_rnn_packed_bytes(const td_rnn_config *cfg) {
    TdRnnGeom g;
    if (td_rnn_geom(cfg, &g) != TD_OK) return 0;
    return g.packed_bytes;
}

// wide layers (rnn_wide.hip) keep the recurrent state of a step in a scratch region, one per direction
// ... and the cluster kernel (rnn_cluster.hip) exchanges h_t between its workgroups through one (twice the size)
static size_t td_state_dir_bytes(const TdRnnGeom &g, int batch) {
    const int nbb = td_cdiv(batch, TD_WAVE_BATCH);
    const size_t wide = g.nu >= 2 ? (size_t)nbb * g.nu * 1024 * sizeof(float) : 0;
    const size_t cl = g.nu == TD_CLUSTER_NU ? td_cluster_xch_dir_bytes(g, nbb) : 0;
    return wide > cl ? wide : cl;
}

// The cluster-persistent kernel takes the small batches of the widths it is instantiated for: NU workgroups per batch
// block and direction, all resident at once.  TD_LAYER_KERNEL_CLUSTER forces it (an error where it cannot run),
// BATCH / UNITS exclude it.
// AUTO: up to this many batch blocks (measured switch point against the other two families)
static int td_cluster_auto_max_blocks(const TdRnnGeom &g) { return TD_CLUSTER_AUTO_BLOCKS / g.dirs; }
static bool td_use_cluster(const TdRnnGeom &g, int nbb, int L) {
    const int force = td_layer_kernel_mode();
    if (force == TD_LAYER_KERNEL_BATCH || force == TD_LAYER_KERNEL_UNITS) return false;
    if (!td_cluster_supported(g, nbb, L)) return false;
    if (force == TD_LAYER_KERNEL_CLUSTER) return true;
    if (g.nu <= TDC_RESIDENT_MAX_NU) return nbb * g.dirs <= TD_RESIDENT_AUTO_WORKGROUPS;
    return nbb <= td_cluster_auto_max_blocks(g);
}

// Which recurrent-layer kernel runs a layer.  rnn_layer.h (a wave = a batch block, all units) needs
// ceil(nbb / 4) * dirs > 128 workgroups to fill the chip, i.e. thousands of sequences; below that -- the
// reference's own batch sizes: 32 for training, 100 for sampling -- rnn_wide.hip's decomposition (the four waves of
// a workgroup split the units of one batch block) puts 4x the waves on a batch block and cuts the per-step
// latency accordingly.  Always rnn_wide.hip above H = 256.
static bool td_use_wide(const TdRnnGeom &g, int nbb) {
    if (g.nu > 8) return true;
    // explicit A/B switch (td_set_layer_kernel): tests pin each kernel family, measurements compare them
    const int force = td_layer_kernel_mode();
    if (force == TD_LAYER_KERNEL_BATCH) return false;
    if (force == TD_LAYER_KERNEL_UNITS) return g.nu >= 2;
    // widths of 2, 3 or 6 unit blocks (HiRID's 36, MIMIC-III's 92, ...): one block per wave, single-block workgroups
    // only, so just for the small batches
    if (g.nu % 4 != 0) return g.nu >= 2 && nbb * g.dirs <= 64;
    // measured at H = 256: the unit-split kernel wins while its workgroups (one or two batch
    // blocks each) fit the chip in one round, i.e. up to 8192 sequences bidirectional
    return td_cdiv(nbb, TD_BLOCK_WAVES) * g.dirs <= 128;
}

extern "C" int td_rnn_layer_kernel_choice(const td_rnn_config *cfg, int32_t batch, int32_t seq_len) {
    TdRnnGeom g;
    if (td_rnn_geom(cfg, &g) != TD_OK || batch <= 0 || seq_len <= 0) return -1;
    const int nbb = td_cdiv(batch, TD_WAVE_BATCH);
    if (td_use_cluster(g, nbb, seq_len)) return TD_LAYER_KERNEL_CLUSTER;
    if (td_layer_kernel_mode() == TD_LAYER_KERNEL_CLUSTER) return -1;
    return td_use_wide(g, nbb) ? TD_LAYER_KERNEL_UNITS : TD_LAYER_KERNEL_BATCH;
}

// 1 where the unit-split family would run this launch as one launch per time step (td_set_step_launch), else 0
extern "C" int td_rnn_step_launch_choice(const td_rnn_config *cfg, int32_t batch, int32_t seq_len) {
    TdRnnGeom g;
    if (td_rnn_geom(cfg, &g) != TD_OK || batch <= 0 || seq_len <= 0) return -1;
    const int nbb = td_cdiv(batch, TD_WAVE_BATCH);
    return !td_use_cluster(g, nbb, seq_len) && td_use_wide(g, nbb) && td_use_step_launch(g.nu, nbb, g.dirs) ? 1 : 0;
}

// workspace: [x0 fragments][act A][act B][wide layers: state, dirs regions]
static void td_ws_layout(const TdRnnGeom &g, int batch, int L, size_t *x0_bytes, size_t *act_bytes) {
    size_t nbb = (size_t)td_cdiv(batch, TD_WAVE_BATCH);
    *x0_bytes = (size_t)L * nbb * g.g0.xfr * TD_FRAG_BYTES;
    *act_bytes = (size_t)L * nbb * (g.dirs * g.kh) * TD_FRAG_BYTES;
}

// Fused head (the batch-block family's top layer, one FiLM row shared by the batch: the sampler): the top layer's weight
// stream with the folded head fragments behind it, and the per-(step, block, direction) partials the layer kernel leaves
// instead of its activations.  Both live behind the state scratch in the workspace.
struct TdHeadFuse {
    size_t stream_off, part_off, total;   // byte offsets in the workspace, and the workspace size with them
    size_t hdir_bytes;                    // one direction of the extended stream
    int hp_stride;                        // floats per lane of the partials
    bool possible;                        // the width and depth have the kernel at all
};
static TdHeadFuse td_head_fuse_layout(const td_rnn_config *cfg, const TdRnnGeom &g, int batch, int L) {
    TdHeadFuse h;
    size_t x0, act;
    td_ws_layout(g, batch, L, &x0, &act);
    const size_t base = (x0 + 2 * act + g.dirs * td_state_dir_bytes(g, batch) + 255) & ~(size_t)255;
    h.possible = g.nu <= 8 && cfg->layers >= 2;
    h.hp_stride = cfg->c_out <= 16 ? 12 : 20;
    const size_t nfr = (size_t)g.g1.nf + 2 * g.kh;
    h.hdir_bytes = ((nfr + TD_STAGE_FRAGS - 1) / TD_STAGE_FRAGS) * TD_STAGE_BYTES;
    h.stream_off = base;
    h.part_off = base + (h.possible ? g.dirs * h.hdir_bytes : 0);
    const size_t nbb = (size_t)td_cdiv(batch, TD_WAVE_BATCH);
    h.total = h.part_off + (h.possible ? (size_t)L * nbb * g.dirs * 64 * h.hp_stride * sizeof(float) : 0);
    return h;
}

extern "C" size_t td_rnn_workspace_bytes(const td_rnn_config *cfg, int32_t batch, int32_t seq_len) {
    TdRnnGeom g;
    if (td_rnn_geom(cfg, &g) != TD_OK || batch <= 0 || seq_len <= 0) return 0;
    return td_head_fuse_layout(cfg, g, batch, seq_len).total;
}

// ---------------------------------------------------------------------------------------------
// weight packing (a derived cache of the torch-layout parameters)
// ---------------------------------------------------------------------------------------------
struct TdPackParams {
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    char *dst;
    int H, In, nu, ng, kh, kx, fu, nf, ns, layer0, gru;
};

// One launch packs every (layer, direction) of a call: blockIdx.y selects the job (22 launches of ~5 us per optimizer
// step were 1.5 % of the reference-batch training step).
#define TD_PACK_JOBS 12
struct TdPackJobs {
    TdPackParams job[TD_PACK_JOBS];
};

__global__ void td_pack_weights_kernel(TdPackJobs jobs) {
    const TdPackParams &p = jobs.job[blockIdx.y];
    const size_t total = (size_t)p.ns * TD_STAGE_FRAGS * 512;
    __bf16 *dst = (__bf16 *)p.dst;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(idx / 512), lane = (int)((idx % 512) / 8), j = (int)(idx % 8);
        const int r = lane & 31, hf = lane >> 5;
        float val = 0.f;
        if (f < p.nf) {
            const int u = f / p.fu, rem0 = f % p.fu;
            const int unit = 32 * u + r;
            int g = 0, part = 0;
            bool lo = false;
            float w = 0.f;
            if (rem0 < 4) {  // bias k-step: column k = 0 carries bf16(b), k = 1 the remainder
                const int slot = rem0;
                float b = 0.f;
                if (unit < p.H) {
                    if (!p.gru || slot < 2) b = p.b_ih[slot * p.H + unit] + p.b_hh[slot * p.H + unit];
                    else if (slot == 2) b = p.b_ih[2 * p.H + unit];
                    else b = p.b_hh[2 * p.H + unit];
                }
                if (hf == 0 && j == 0) w = b;
                if (hf == 0 && j == 1) { w = b; lo = true; }
            } else if (rem0 - 4 < p.kh * p.ng) {  // recurrent k-step
                const int rem = rem0 - 4;
                const int ks = rem / p.ng;
                g = rem % p.ng;
                const int kk = 16 * ks + 8 * (j >> 2) + 4 * hf + (j & 3);
                if (unit < p.H && kk < p.H) w = p.w_hh[(size_t)(g * p.H + unit) * p.H + kk];
            } else if (p.layer0) {
                const int x = rem0 - 4 - p.kh * p.ng;
                part = x & 1;
                const int kx = (x >> 1) / p.ng;
                g = (x >> 1) % p.ng;
                const int ch = 16 * kx + 8 * hf + j;
                if (unit < p.H && ch < p.In) w = p.w_ih[(size_t)(g * p.H + unit) * p.In + ch];
                lo = part == 1;
            } else {
                const int x = rem0 - 4 - p.kh * p.ng;
                const int kx = x / p.ng;
                g = x % p.ng;
                const int din = kx / p.kh, ks = kx % p.kh;
                const int kk = 16 * ks + 8 * (j >> 2) + 4 * hf + (j & 3);
                if (unit < p.H && kk < p.H) w = p.w_ih[(size_t)(g * p.H + unit) * p.In + din * p.H + kk];
            }
            // Gate pre-activations leave the MFMA already multiplied by the factor their nonlinearity's exp2 needs:
            // sigmoid(a) = 1 / (1 + 2^(-log2(e) a)), tanh(a) = 1 - 2 / (2^(2 log2(e) a) + 1).  Folding the factor into
            // the weights and biases before the bf16 rounding saves a multiply per gate value in the layer kernel.
            const bool tanh_gate = (rem0 < 4) ? (p.gru ? rem0 >= 2 : rem0 == 2) : g == 2;
            w *= tanh_gate ? TD_GATE_SCALE_TANH : TD_GATE_SCALE_SIGMOID;
            if (lo) {
                const __bf16 hi = (__bf16)w;
                val = w - (float)hi;
            } else {
                val = w;
            }
        }
        dst[idx] = (__bf16)val;
    }
}

extern "C" int td_rnn_pack_weights(const td_rnn_config *cfg, const float *const *lw, void *packed, void *stream) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(lw && packed, "NULL argument");
    (void)td_async_error_word(true);   // rnn_cluster.h: the word the cluster kernels raise on a timed-out hand-off
    for (int i = 0; i < cfg->layers * cfg->dirs * 4; ++i)
        TD_REQUIRE(lw[i], "NULL weight pointer for layer %d dir %d", i / 4 / cfg->dirs, i / 4 % cfg->dirs);
    TdPackJobs jobs;
    int nj = 0, blocks = 1;
    auto flush = [&]() {
        if (nj) hipLaunchKernelGGL(td_pack_weights_kernel, dim3(blocks, nj), dim3(256), 0, (hipStream_t)stream, jobs);
        nj = 0; blocks = 1;
    };
    for (int l = 0; l < cfg->layers; ++l)
        for (int d = 0; d < cfg->dirs; ++d) {
            const TdLayerGeom &lg = l == 0 ? g.g0 : g.g1;
            const float *const *w = lw + (size_t)(l * cfg->dirs + d) * 4;
            TdPackParams &p = jobs.job[nj++];
            p.w_ih = w[0]; p.w_hh = w[1]; p.b_ih = w[2]; p.b_hh = w[3];
            p.dst = (char *)packed + g.layer_off[l] + (size_t)d * lg.dir_bytes;
            p.H = cfg->hidden; p.In = l == 0 ? cfg->c_in : cfg->dirs * cfg->hidden;
            p.nu = lg.nu; p.ng = lg.ng; p.kh = lg.kh; p.kx = lg.kx; p.fu = lg.fu; p.nf = lg.nf; p.ns = lg.ns;
            p.layer0 = l == 0; p.gru = cfg->gru;
            const size_t total = (size_t)lg.ns * TD_STAGE_FRAGS * 512;
            const int b = (int)((total + 255) / 256);
            blocks = b > blocks ? (b > 4096 ? 4096 : b) : blocks;
            if (nj == TD_PACK_JOBS) { flush(); TD_LAUNCH_CHECK(); }
        }
    flush();
    TD_LAUNCH_CHECK();
    if (g.cl_dir[1] != 0) return td_cluster_pack_all(cfg, g, lw, (char *)packed, (hipStream_t)stream);
    return TD_OK;
}

// ---------------------------------------------------------------------------------------------
// time embedding + FiLM linear (blocks.py:17-24, :324-329, :285).  One block per time value.
// ---------------------------------------------------------------------------------------------
struct TdFilmParams {
    const float *w1, *b1, *w2, *b2, *wf, *bf;
    const int64_t *t;
    float *film, *temb;
    int embed_dim, time_dim, film_dim;
};

__global__ __launch_bounds__(256) void td_time_film_kernel(TdFilmParams p) {
    __shared__ float sinus[1024], h1[1024], sact[1024];
    const int i = blockIdx.x, tid = threadIdx.x;
    const float tv = (float)p.t[i];
    const int half = p.embed_dim / 2;
    const float emb = (float)(-(9.210340371976184 /* ln 1e4 */ / (double)(half - 1)));
    for (int j = tid; j < half; j += 256) {
        const float a = tv * expf((float)j * emb);
        sinus[j] = sinf(a);
        sinus[half + j] = cosf(a);
    }
    __syncthreads();
    for (int r = tid; r < p.time_dim; r += 256) {
        float v = p.b1[r];
        const float *w = p.w1 + (size_t)r * p.embed_dim;
        for (int k = 0; k < p.embed_dim; ++k) v = fmaf(w[k], sinus[k], v);
        h1[r] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    }
    __syncthreads();
    for (int r = tid; r < p.time_dim; r += 256) {
        float v = p.b2[r];
        const float *w = p.w2 + (size_t)r * p.time_dim;
        for (int k = 0; k < p.time_dim; ++k) v = fmaf(w[k], h1[k], v);
        if (p.temb) p.temb[(size_t)i * p.time_dim + r] = v;
        sact[r] = v / (1.0f + expf(-v));
    }
    __syncthreads();
    for (int r = tid; r < p.film_dim; r += 256) {
        float v = p.bf[r];
        const float *w = p.wf + (size_t)r * p.time_dim;
        for (int k = 0; k < p.time_dim; ++k) v = fmaf(w[k], sact[k], v);
        p.film[(size_t)i * p.film_dim + r] = v;
    }
}

extern "C" int td_rnn_time_film(const td_rnn_config *cfg, const float *const *misc, const int64_t *t, int32_t n_t,
                                float *film, float *temb_out, void *stream) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(misc && t && film && n_t > 0, "bad argument");
    TdFilmParams p;
    p.w1 = misc[TD_TE_W1]; p.b1 = misc[TD_TE_B1]; p.w2 = misc[TD_TE_W2]; p.b2 = misc[TD_TE_B2];
    p.wf = misc[TD_FILM_W]; p.bf = misc[TD_FILM_B];
    p.t = t; p.film = film; p.temb = temb_out;
    p.embed_dim = cfg->embed_dim; p.time_dim = cfg->time_dim; p.film_dim = 2 * cfg->dirs * cfg->hidden;
    hipLaunchKernelGGL(td_time_film_kernel, dim3(n_t), dim3(256), 0, (hipStream_t)stream, p);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

// Backward of the same path (autograd through blocks.py:324-329 and :285), given d film (n_t, 2 D H).  Stage 1, one block
// per row of t: recompute the forward values (cheaper than saving them), back-propagate the activations, leave per
// row the vectors the weight gradients are outer products of, add the bias gradients.  Stage 2: every weight gradient is
// a sum over the rows of an outer product; one thread per weight element walks the rows.
struct TdFilmBwdParams {
    const float *w1, *b1, *w2, *b2, *wf, *bf;
    const int64_t *t;
    const float *gfilm;
    float *rows;                       // per row: [sinus (E) | a1 (T) | s (T) | d v1 (T) | d v2 (T)]
    float *d_w1, *d_b1, *d_w2, *d_b2, *d_wf, *d_bf;   // accumulated
    int embed_dim, time_dim, film_dim, n_t;
};

// 1024 threads: the long dot products (d s over the 2 D H FiLM rows, d a1 over the time_dim rows) are cut into
// blockDim / time_dim parts per output element and added up through LDS -- with one thread per element and a few dozen
// rows the kernel was a 1 000-iteration dependent loop on 64 CUs (0.37 ms of a 7.6 ms training step at the reference's batch)
#define TD_FILM_BWD_THREADS 1024
__global__ __launch_bounds__(TD_FILM_BWD_THREADS) void td_time_film_bwd_rows_kernel(TdFilmBwdParams p) {
    __shared__ float sinus[1024], v1[1024], a1[1024], v2[1024], tmp[1024], part[TD_FILM_BWD_THREADS];
    const int i = blockIdx.x, tid = threadIdx.x, nth = blockDim.x;
    const int E = p.embed_dim, T = p.time_dim, F = p.film_dim;
    const float tv = (float)p.t[i];
    const int half = E / 2;
    const float emb = (float)(-(9.210340371976184 / (double)(half - 1)));
    for (int j = tid; j < half; j += nth) {
        const float a = tv * expf((float)j * emb);
        sinus[j] = sinf(a);
        sinus[half + j] = cosf(a);
    }
    __syncthreads();
    // forward values, as td_time_film_kernel computes them
    for (int r = tid; r < T; r += nth) {
        float v = p.b1[r];
        const float *w = p.w1 + (size_t)r * E;
        for (int k = 0; k < E; ++k) v = fmaf(w[k], sinus[k], v);
        v1[r] = v;
        a1[r] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    }
    __syncthreads();
    for (int r = tid; r < T; r += nth) {
        float v = p.b2[r];
        const float *w = p.w2 + (size_t)r * T;
        for (int k = 0; k < T; ++k) v = fmaf(w[k], a1[k], v);
        v2[r] = v;
    }
    __syncthreads();
    const float *gf = p.gfilm + (size_t)i * F;
    float *row = p.rows + (size_t)i * (E + 4 * T);
    // parts of a column sum: thread (k, q) adds rows q, q + nq, ... (T <= blockDim: checked by the launcher)
    const int nq = nth / T, k = tid % T, q = tid / T;
    // d s = W_f^T d film ; d v2 = d s * silu'(v2)
    {
        float ds = 0.f;
        if (q < nq)
            for (int r = q; r < F; r += nq) ds = fmaf(p.wf[(size_t)r * T + k], gf[r], ds);
        part[tid] = ds;
    }
    __syncthreads();
    if (tid < T) {
        float ds = 0.f;
        for (int qq = 0; qq < nq; ++qq) ds += part[qq * T + tid];
        const float v = v2[tid], sg = 1.0f / (1.0f + expf(-v));
        const float dv = ds * sg * (1.0f + v * (1.0f - sg));
        tmp[tid] = dv;
        row[E + T + tid] = v * sg;          // s = silu(v2)
        row[E + 3 * T + tid] = dv;
        atomicAdd(&p.d_b2[tid], dv);
    }
    for (int r = tid; r < F; r += nth) atomicAdd(&p.d_bf[r], gf[r]);
    __syncthreads();
    // d a1 = W_2^T d v2 ; d v1 = d a1 * gelu'(v1)
    {
        float da = 0.f;
        if (q < nq)
            for (int r = q; r < T; r += nq) da = fmaf(p.w2[(size_t)r * T + k], tmp[r], da);
        part[tid] = da;
    }
    __syncthreads();
    if (tid < T) {
        float da = 0.f;
        for (int qq = 0; qq < nq; ++qq) da += part[qq * T + tid];
        const float v = v1[tid];
        const float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f)), pdf = 0.3989422804014327f * expf(-0.5f * v * v);
        const float dv = da * (cdf + v * pdf);
        row[E + tid] = a1[tid];
        row[E + 2 * T + tid] = dv;
        atomicAdd(&p.d_b1[tid], dv);
    }
    for (int kk = tid; kk < E; kk += nth) row[kk] = sinus[kk];
}

// d W[r][k] += sum_i left[i][r] * right[i][k]   (left / right: columns of the per-row records, or d film; fp32)
// A workgroup owns a 64 x 64 tile of d W (a thread 4 x 4 of it) and walks the rows i in stages of 32 through LDS; a tile
// is summed in row order by one workgroup (from 256 rows up several, which add with float atomics).  The weight gradients of the
// time-embedding MLP and the FiLM linear: at most 1024 x 256 outputs over `batch` rows, 1 GFLOP at batch 2048.
#define TD_OS_TILE 64
#define TD_OS_ROWS 32
__global__ __launch_bounds__(256) void td_outer_sum_kernel(const float *__restrict__ left, size_t lstride, const float *__restrict__ right,
                                                            size_t rstride, float *__restrict__ dw, int R, int K, int n) {
    __shared__ float sl[TD_OS_ROWS][TD_OS_TILE + 4], sr[TD_OS_ROWS][TD_OS_TILE + 4];
    const int r0 = blockIdx.x * TD_OS_TILE, k0 = blockIdx.y * TD_OS_TILE;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;
    float acc[4][4] = {};
    // gridDim.z > 1: the rows are cut into chunks that add into d W with float atomics (large batches only)
    const int per = TD_OS_ROWS * ((n + TD_OS_ROWS * (int)gridDim.z - 1) / (TD_OS_ROWS * (int)gridDim.z));
    const int ibeg = blockIdx.z * per, iend = min(n, ibeg + per);
    for (int i0 = ibeg; i0 < iend; i0 += TD_OS_ROWS) {
        for (int e = tid; e < TD_OS_ROWS * TD_OS_TILE; e += 256) {
            const int ii = e / TD_OS_TILE, cc = e % TD_OS_TILE;
            const bool row = i0 + ii < iend;
            sl[ii][cc] = (row && r0 + cc < R) ? left[(size_t)(i0 + ii) * lstride + r0 + cc] : 0.f;
            sr[ii][cc] = (row && k0 + cc < K) ? right[(size_t)(i0 + ii) * rstride + k0 + cc] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int ii = 0; ii < TD_OS_ROWS; ++ii) {
            const f32x4 a = *(const f32x4 *)&sl[ii][4 * ty], b = *(const f32x4 *)&sr[ii][4 * tx];
#pragma unroll
            for (int y = 0; y < 4; ++y)
#pragma unroll
                for (int x = 0; x < 4; ++x) acc[y][x] = fmaf(a[y], b[x], acc[y][x]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int y = 0; y < 4; ++y)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int r = r0 + 4 * ty + y, k = k0 + 4 * tx + x;
            if (r < R && k < K) {
                if (gridDim.z > 1) atomicAdd(&dw[(size_t)r * K + k], acc[y][x]);
                else dw[(size_t)r * K + k] += acc[y][x];
            }
        }
}

extern "C" size_t td_rnn_time_film_backward_workspace_bytes(const td_rnn_config *cfg, int32_t n_t) {
    if (!cfg || n_t <= 0) return 0;
    return (size_t)n_t * (cfg->embed_dim + 4 * cfg->time_dim) * sizeof(float);
}

extern "C" int td_rnn_time_film_backward(const td_rnn_config *cfg, const float *const *misc, const int64_t *t, int32_t n_t,
                                         const float *grad_film, float *const *grad_misc, void *workspace, void *stream_) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(misc && t && grad_film && grad_misc && workspace && n_t > 0, "bad argument");
    hipStream_t stream = (hipStream_t)stream_;
    TdFilmBwdParams p;
    p.w1 = misc[TD_TE_W1]; p.b1 = misc[TD_TE_B1]; p.w2 = misc[TD_TE_W2]; p.b2 = misc[TD_TE_B2];
    p.wf = misc[TD_FILM_W]; p.bf = misc[TD_FILM_B];
    p.t = t; p.gfilm = grad_film; p.rows = (float *)workspace;
    p.d_w1 = grad_misc[TD_TE_W1]; p.d_b1 = grad_misc[TD_TE_B1]; p.d_w2 = grad_misc[TD_TE_W2]; p.d_b2 = grad_misc[TD_TE_B2];
    p.d_wf = grad_misc[TD_FILM_W]; p.d_bf = grad_misc[TD_FILM_B];
    TD_REQUIRE(p.d_w1 && p.d_b1 && p.d_w2 && p.d_b2 && p.d_wf && p.d_bf, "NULL gradient pointer for the time-embedding parameters");
    const int E = cfg->embed_dim, T = cfg->time_dim, F = 2 * cfg->dirs * cfg->hidden;
    p.embed_dim = E; p.time_dim = T; p.film_dim = F; p.n_t = n_t;
    TD_REQUIRE(T <= 1024 && E <= 1024, "time_dim / embed_dim above 1024");
    hipLaunchKernelGGL(td_time_film_bwd_rows_kernel, dim3(n_t), dim3(TD_FILM_BWD_THREADS), 0, stream, p);
    TD_LAUNCH_CHECK();
    const size_t rs = (size_t)E + 4 * T;
    const float *rows = p.rows;
    // d W_f = d film^T s ; d W_2 = d v2^T a1 ; d W_1 = d v1^T sinus: sums over the rows of outer products
    auto outer = [&](const float *left, size_t ls, const float *right, float *dw, int R, int K) -> int {
        const int tiles = td_cdiv(R, TD_OS_TILE) * td_cdiv(K, TD_OS_TILE);
        int nz = 1;
        while (tiles * nz < 512 && n_t / (2 * nz) >= 128) nz *= 2;
        hipLaunchKernelGGL(td_outer_sum_kernel, dim3(td_cdiv(R, TD_OS_TILE), td_cdiv(K, TD_OS_TILE), nz), dim3(256), 0, stream, left, ls,
                           right, rs, dw, R, K, n_t);
        TD_LAUNCH_CHECK();
        return TD_OK;
    };
    if ((rc = outer(grad_film, (size_t)F, rows + E + T, p.d_wf, F, T)) != TD_OK) return rc;
    if ((rc = outer(rows + E + 3 * T, rs, rows + E, p.d_w2, T, T)) != TD_OK) return rc;
    return outer(rows + E + 2 * T, rs, rows, p.d_w1, T, E);
}

// ---------------------------------------------------------------------------------------------
// layer-0 input: (B, C, L) fp32 -> B-operand fragments, split into bf16 hi + lo parts so the first
// input projection keeps ~16 mantissa bits (the one-hot logs are {0, -69.0776}; plain bf16 would
// round them to -69.0).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void td_pack_input_kernel(const float *__restrict__ x, char *__restrict__ dst, int B,
                                                            int C, int L, int nbb, int kx0) {
    const size_t total = (size_t)L * nbb * kx0 * 64;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int lane = (int)(idx % 64);
        const int kx = (int)((idx / 64) % kx0);
        const int bblk = (int)((idx / (64 * (size_t)kx0)) % nbb);
        const int l = (int)(idx / (64 * (size_t)kx0 * nbb));
        const int b = bblk * 32 + (lane & 31), hf = lane >> 5;
        bf16x8 hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ch = 16 * kx + 8 * hf + j;
            float v = 0.f;
            if (b < B && ch < C) v = x[((size_t)b * C + ch) * L + l];
            const __bf16 h = (__bf16)v;
            hi[j] = h;
            lo[j] = (__bf16)(v - (float)h);
        }
        char *o = dst + (((size_t)l * nbb + bblk) * (2 * kx0) + 2 * kx) * TD_FRAG_BYTES + lane * 16;
        *(bf16x8 *)o = hi;
        *(bf16x8 *)(o + TD_FRAG_BYTES) = lo;
    }
}

// ---------------------------------------------------------------------------------------------
// sampler: diffusion state -> layer-0 input fragments in one pass (= td_build_input_kernel followed by
// td_pack_input_kernel, bit for bit).  The state is channel-major with l contiguous, the fragments want 32
// sequences x 8 channels per 1 KiB: a (channels x 32 sequences x 16 positions) tile goes through LDS so that both
// the reads (64 B runs along l) and the writes (1 KiB fragments) are coalesced.
// ---------------------------------------------------------------------------------------------
#define TD_SF_LT 16
__global__ __launch_bounds__(256) void td_state_to_frags_kernel(td_diffusion_config c, int sumk, const float *__restrict__ z,
                                                                const uint8_t *__restrict__ idx, char *__restrict__ dst, int B,
                                                                int nbb, int kx0) {
    extern __shared__ float tile[];   // [16 kx0][32][TD_SF_LT + 1]: 34 KiB at c_in <= 16, four workgroups per CU
    const int L = c.seq_len, CP = 16 * kx0, C_io = c.n_num + sumk;
    const int l0 = blockIdx.x * TD_SF_LT, bblk = blockIdx.y, tid = threadIdx.x;
    const int RS = TD_SF_LT + 1;
    for (int e = tid; e < (CP - C_io) * 32 * TD_SF_LT; e += 256) {   // channels past the denoiser width
        const int ll = e % TD_SF_LT, bb = (e / TD_SF_LT) % 32, ch = C_io + e / (TD_SF_LT * 32);
        tile[(ch * 32 + bb) * RS + ll] = 0.f;
    }
    for (int e = tid; e < c.n_num * 32 * TD_SF_LT; e += 256) {
        const int ll = e % TD_SF_LT, bb = (e / TD_SF_LT) % 32, ch = e / (TD_SF_LT * 32);
        const int b = bblk * 32 + bb, l = l0 + ll;
        tile[(ch * 32 + bb) * RS + ll] = (b < B && l < L) ? z[((size_t)b * c.n_num + ch) * L + l] : 0.f;
    }
    for (int e = tid; e < c.n_cat * 32 * TD_SF_LT; e += 256) {
        const int ll = e % TD_SF_LT, bb = (e / TD_SF_LT) % 32, v = e / (TD_SF_LT * 32);
        const int b = bblk * 32 + bb, l = l0 + ll;
        const bool ok = b < B && l < L;
        const int cur = ok ? idx[((size_t)b * c.n_cat + v) * L + l] : -1;
        int off = c.n_num;
        for (int w = 0; w < v; ++w) off += c.num_classes[w];
        for (int k = 0; k < c.num_classes[v]; ++k)
            tile[((off + k) * 32 + bb) * RS + ll] = !ok ? 0.f : (k == cur ? 0.f : TD_LOG_1E30);   // mixed_diffusion.py:44-51
    }
    __syncthreads();
    const int wave = tid >> 6, lane = tid & 63, m = lane & 31, hf = lane >> 5;
    for (int ll = wave; ll < TD_SF_LT; ll += 4) {
        const int l = l0 + ll;
        if (l >= L) break;
        for (int kx = 0; kx < kx0; ++kx) {
            bf16x8 hi, lo;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float v = tile[((16 * kx + 8 * hf + j) * 32 + m) * RS + ll];
                const __bf16 h = (__bf16)v;
                hi[j] = h;
                lo[j] = (__bf16)(v - (float)h);
            }
            char *o = dst + (((size_t)l * nbb + bblk) * (2 * kx0) + 2 * kx) * TD_FRAG_BYTES + lane * 16;
            *(bf16x8 *)o = hi;
            *(bf16x8 *)(o + TD_FRAG_BYTES) = lo;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// head: LayerNorm(D*H) -> FiLM -> fc -> (B, C_out, L)   (blocks.py:293-297, :340-341)
//
// One wave per (position, batch block of 32 sequences).  The last layer's fragments of a
// position are read ONCE into registers (KT x 16 B per lane); LayerNorm statistics are two register
// passes (lanes m and m + 32 hold the two halves of sequence m); the normalised + FiLMed features stay in
// fragment layout, which is exactly the B operand of v_mfma_f32_32x32x16_bf16, so the projection is
//     y^T[32 (channels, padded) x 32 sequences] += fc_w[32 x 16] x feat[16 x 32]
// with fc_w packed into A fragments in LDS once per workgroup.  Features and weights are both split into
// bf16 hi + lo parts and the three leading products are kept, so the projection is fp32-accurate.
// HBM traffic: the activations once (L*D*H*2 B per sequence) + the output.
// ---------------------------------------------------------------------------------------------
struct TdHeadParams {
    const char *act;  // last layer's fragments
    const float *ln_w, *ln_b, *fc_w, *fc_b, *film;
    const float *ffr;   // MODE 2: per-sequence FiLM in fragment order (td_film_frag_kernel)
    const int32_t *film_row;
    float *y;
    int film_rows, B, L, nbb, H, hpad, kh, dirs, c_out;
};

// Per-sequence FiLM rows (training: every sequence has its own time step) folded with the LayerNorm affine and laid out
// the way the head kernels read them: [batch block][fragment ks][a | b][lane][8] fp32 with
//     a = ln_w * (1 + scale_b),   b = ln_b * (1 + scale_b) + shift_b        (0 for padded units and sequences)
// so a lane's multipliers for a fragment are two 32-byte vectors instead of 16 scattered words of the (B, 2 D H) array,
// read again at every position.
__global__ __launch_bounds__(256) void td_film_frag_kernel(const float *__restrict__ film, const float *__restrict__ ln_w,
                                                           const float *__restrict__ ln_b, float *__restrict__ ffr, int B, int nbb, int H,
                                                           int kh, int dirs) {
    const int KT = dirs * kh, DH = dirs * H;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)nbb * KT * 64) return;
    const int lane = (int)(idx % 64), ks = (int)((idx / 64) % KT), bblk = (int)(idx / ((size_t)64 * KT));
    const int m = lane & 31, hf = lane >> 5, b = bblk * 32 + m;
    float a[8], c[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int unit = 16 * (ks % kh) + 8 * (j >> 2) + 4 * hf + (j & 3);
        const bool ok = unit < H && b < B;
        const int feat = (ks / kh) * H + (ok ? unit : 0);
        const float sc = ok ? film[(size_t)b * 2 * DH + feat] + 1.0f : 0.f, sh = ok ? film[(size_t)b * 2 * DH + DH + feat] : 0.f;
        // the same roundings as td_head_kernel's MODE 1 (one multiply, one fused multiply-add)
        a[j] = ok ? __fmul_rn(ln_w[feat], sc) : 0.f;
        c[j] = ok ? __fmaf_rn(ln_b[feat], sc, sh) : 0.f;
    }
    float *o = ffr + (((size_t)bblk * KT + ks) * 2) * 512 + lane * 8;
    *(f32x4 *)o = f32x4{a[0], a[1], a[2], a[3]};
    *(f32x4 *)(o + 4) = f32x4{a[4], a[5], a[6], a[7]};
    *(f32x4 *)(o + 512) = f32x4{c[0], c[1], c[2], c[3]};
    *(f32x4 *)(o + 516) = f32x4{c[4], c[5], c[6], c[7]};
}

// MODE 0: one FiLM row for the whole batch (the sampler); 1: one row per sequence, read from the (B, 2 D H) array;
// 2: one row per sequence, read from the fragment-ordered copy above (the training forward)
template <int KT, int MODE>
__global__ __launch_bounds__(256, 2) void td_head_kernel(TdHeadParams p) {
    constexpr bool PER_SEQ = MODE != 0;
    // [KT][hi|lo] fc_w A-fragments, then three [KT*16] fp32 vectors in fragment order (k = 8 hf + j)
    __shared__ __attribute__((aligned(16))) char hl[KT * 2 * TD_FRAG_BYTES + 3 * KT * 16 * sizeof(float)];
    char *wfr = hl;
    float *v_a = (float *)(hl + (size_t)KT * 2 * TD_FRAG_BYTES), *v_b = v_a + KT * 16, *v_msk = v_b + KT * 16;
    const int DH = p.dirs * p.H;
    // PER_SEQ (training): every sequence has its own time step, i.e. its own FiLM row
    const float *film0 = p.film + (size_t)(p.film_row ? p.film_row[0] : 0) * 2 * DH;

    for (int idx = threadIdx.x; idx < KT * 64 * 8; idx += 256) {
        const int ks = idx / 512, ln = (idx % 512) / 8, j = idx % 8;
        const int c = ln & 31, hf = ln >> 5;
        const int d = ks / p.kh, unit = 16 * (ks % p.kh) + 8 * (j >> 2) + 4 * hf + (j & 3);
        const float w = (c < p.c_out && unit < p.H) ? p.fc_w[(size_t)c * DH + d * p.H + unit] : 0.f;
        const __bf16 hi = (__bf16)w;
        ((__bf16 *)(wfr + (size_t)(2 * ks) * TD_FRAG_BYTES))[ln * 8 + j] = hi;
        ((__bf16 *)(wfr + (size_t)(2 * ks + 1) * TD_FRAG_BYTES))[ln * 8 + j] = (__bf16)(w - (float)hi);
    }
    for (int idx = threadIdx.x; idx < KT * 16; idx += 256) {
        const int ks = idx / 16, hf = (idx % 16) / 8, j = idx % 8;
        const int d = ks / p.kh, unit = 16 * (ks % p.kh) + 8 * (j >> 2) + 4 * hf + (j & 3);
        const bool ok = unit < p.H;
        const int feat = d * p.H + (ok ? unit : 0);
        const float lw = ok ? p.ln_w[feat] : 0.f, lb = ok ? p.ln_b[feat] : 0.f;
        if (PER_SEQ) {
            v_a[idx] = lw; v_b[idx] = lb;
        } else {  // y = xhat * ln_w * (scale + 1) + ln_b * (scale + 1) + shift
            const float sc = ok ? film0[feat] + 1.0f : 0.f, sh = ok ? film0[DH + feat] : 0.f;
            v_a[idx] = lw * sc; v_b[idx] = lb * sc + sh;
        }
        v_msk[idx] = ok ? 1.f : 0.f;
    }
    __syncthreads();

    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, hf = lane >> 5;
    // MODE 2: the batch blocks are dealt to the XCDs (workgroups go round-robin by index), so a block's 4 KT KiB of FiLM
    // fragments are fetched into ONE L2 and re-read there at every position
    const int xcd = MODE == 2 ? blockIdx.x % 8 : 0, xstep = MODE == 2 ? 8 : 1;
    const int nbx = MODE == 2 ? (p.nbb - xcd + 7) / 8 : p.nbb;            // batch blocks this workgroup draws from
    const size_t ntask = (size_t)p.L * nbx;
    const size_t first = (size_t)(blockIdx.x / xstep) * 4 + wave, stride = (size_t)(gridDim.x / xstep) * 4;
#pragma unroll 1
    for (size_t tx = first; tx < ntask; tx += stride) {
        const int l = (int)(tx / nbx), bblk = xcd + xstep * (int)(tx % nbx);
        const size_t task = (size_t)l * p.nbb + bblk;
        const int b = bblk * 32 + m;
        const float *filmb = p.film + (size_t)(b < p.B ? b : 0) * 2 * DH;
        const bf16x8 *fp = (const bf16x8 *)(p.act + task * KT * TD_FRAG_BYTES) + lane;
        // the per-feature vectors are re-read from LDS for every task: hoisted out of the task loop they would
        // need 24 * KT registers per lane
        const float *va = v_a + hf * 8, *vb = v_b + hf * 8, *vm = v_msk + hf * 8;
        const char *wf = wfr + lane * 16;
        asm volatile("" : "+v"(va), "+v"(vb), "+v"(vm), "+v"(wf));
        bf16x8 v[KT];
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) v[ks] = __builtin_nontemporal_load(fp + ks * 64);
        float sum = 0.f;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += (float)v[ks][j];  // padded units hold exact zeros
        sum += __shfl_xor(sum, 32);
        const float mu = sum / (float)DH;
        float sq = 0.f;
        // each pass unpacks the bf16 values again: kept unpacked across passes they would need 8 * KT registers
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) asm volatile("" : "+v"(v[ks]));
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            const f32x4 m0 = *(const f32x4 *)(vm + ks * 16), m1 = *(const f32x4 *)(vm + ks * 16 + 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dlt = ((float)v[ks][j] - mu) * (j < 4 ? m0[j & 3] : m1[j & 3]);
                sq = fmaf(dlt, dlt, sq);
            }
        }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq / (float)DH + 1e-5f);
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) asm volatile("" : "+v"(v[ks]));
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            f32x4 a0, a1, b0, b1;
            if constexpr (MODE == 2) {
                const f32x4 *fa = (const f32x4 *)(p.ffr + (((size_t)bblk * KT + ks) * 2) * 512) + lane * 2;
                a0 = fa[0]; a1 = fa[1]; b0 = fa[128]; b1 = fa[129];
            } else {
                a0 = *(const f32x4 *)(va + ks * 16); a1 = *(const f32x4 *)(va + ks * 16 + 4);
                b0 = *(const f32x4 *)(vb + ks * 16); b1 = *(const f32x4 *)(vb + ks * 16 + 4);
            }
            bf16x8 yh, yl;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float av = j < 4 ? a0[j & 3] : a1[j & 3], bv = j < 4 ? b0[j & 3] : b1[j & 3];
                if constexpr (MODE == 1) {
                    const int unit = 16 * (ks % p.kh) + 8 * (j >> 2) + 4 * hf + (j & 3);
                    const int feat = (ks / p.kh) * p.H + (unit < p.H ? unit : 0);
                    const float sc = filmb[feat] + 1.0f, sh = filmb[DH + feat];  // padded units: av = bv = 0, mask below
                    const float mk = vm[ks * 16 + j];
                    bv = mk != 0.f ? __fmaf_rn(bv, sc, sh) : 0.f;
                    av = __fmul_rn(av, sc);
                }
                // pinned roundings: every MODE gives the same bits for the same multipliers
                const float yv = __fmaf_rn(__fmul_rn((float)v[ks][j] - mu, rstd), av, bv);
                const __bf16 h = (__bf16)yv;
                yh[j] = h;
                yl[j] = (__bf16)(yv - (float)h);
            }
            const bf16x8 wh = *(const bf16x8 *)(wf + (size_t)(2 * ks) * TD_FRAG_BYTES);
            const bf16x8 wl = *(const bf16x8 *)(wf + (size_t)(2 * ks + 1) * TD_FRAG_BYTES);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, yh, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, yl, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, yh, acc, 0, 0, 0);
        }
        // accumulator row r of lane (m, hf) = channel (r&3) + 8(r>>2) + 4hf of sequence m
        if (b < p.B) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * hf;
                if (c < p.c_out) p.y[((size_t)b * p.c_out + c) * p.L + l] = acc[r] + p.fc_b[c];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// fused head (sampler): y = fc(FiLM(LayerNorm(h))) with ONE FiLM row for the whole batch is linear in h up to the
// LayerNorm statistics:
//     y[c] = rstd * ( sum_f w'[c][f] h[f]  -  mu * sum_f w'[c][f] )  +  sum_f fc_w[c][f] (ln_b[f] (1 + scale[f]) + shift[f])  +  fc_b[c]
//     w'[c][f] = fc_w[c][f] * ln_w[f] * (1 + scale[f])
// The top layer's kernel (rnn_layer.h, HEAD) multiplies w' -- as bf16 hi + lo A fragments appended to its weight stream --
// with the h_t fragments it has just produced and leaves, per direction, the partial projection and sum h, sum h^2;
// td_head_combine_kernel adds the directions.  The layer's activations (4.5 GB per reverse step at the bench shape) are
// neither written nor read.
// ---------------------------------------------------------------------------------------------
struct TdHeadFoldParams {
    const char *src;          // the top layer's packed stream, direction 0
    size_t src_dir;
    char *dst;                // extended stream, direction 0
    size_t dst_dir;
    const float *ln_w, *fc_w, *film;
    const int32_t *film_row;
    int nfb, H, kh, dirs, c_out;
};

__global__ __launch_bounds__(256) void td_head_fold_kernel(TdHeadFoldParams p) {
    const int DH = p.dirs * p.H;
    const float *film = p.film + (size_t)(p.film_row ? p.film_row[0] : 0) * 2 * DH;
    // the unit blocks' fragments, unchanged (16 bytes per thread)
    const size_t nvec = (size_t)p.dirs * p.nfb * 64;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const int d = (int)(i / ((size_t)p.nfb * 64));
        const size_t r = i % ((size_t)p.nfb * 64);
        ((bf16x8 *)(p.dst + (size_t)d * p.dst_dir))[r] = ((const bf16x8 *)(p.src + (size_t)d * p.src_dir))[r];
    }
    // the head fragments: [k-step][hi | lo], A operand: lane (row = channel, hf), element j <-> unit of B's k index 8 hf + j
    const size_t nh = (size_t)p.dirs * 2 * p.kh * 512;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nh; i += (size_t)gridDim.x * 256) {
        const int j = (int)(i % 8), ln = (int)((i / 8) % 64), f = (int)((i / 512) % (2 * p.kh)), d = (int)(i / ((size_t)512 * 2 * p.kh));
        const int c = ln & 31, hf = ln >> 5, ks = f >> 1;
        const int unit = 16 * ks + 8 * (j >> 2) + 4 * hf + (j & 3);
        float w = 0.f;
        if (c < p.c_out && unit < p.H) {
            const int feat = d * p.H + unit;
            w = p.fc_w[(size_t)c * DH + feat] * p.ln_w[feat] * (film[feat] + 1.0f);
        }
        const __bf16 hi = (__bf16)w;
        ((__bf16 *)(p.dst + (size_t)d * p.dst_dir + ((size_t)p.nfb + f) * TD_FRAG_BYTES))[ln * 8 + j] = (f & 1) ? (__bf16)(w - (float)hi) : hi;
    }
}

__device__ __forceinline__ int td_cdiv_dev(int a, int b) { return (a + b - 1) / b; }

struct TdHeadCombineParams {
    const float *hpart;
    const float *ln_w, *ln_b, *fc_w, *fc_b, *film;
    const int32_t *film_row;
    float *y;
    int hp_stride, B, L, nbb, H, dirs, c_out;
};

// A workgroup takes one batch block and TD_HC_STEPS consecutive steps (a wave a quarter of them): the results go through
// an LDS tile [channel][sequence][step] and leave as runs of TD_HC_STEPS consecutive floats of a (sequence, channel) row of
// y (B, C_out, L) -- written one value at a time at a stride of L floats, as the first version did, the kernel took
// 0.91 ms for 1.1 GB.
#define TD_HC_STEPS 16
__global__ __launch_bounds__(256) void td_head_combine_kernel(TdHeadCombineParams p) {
    __shared__ float s_wsum[32], s_cst[32];
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [c_out][32][16 + 4] floats: rows 16-byte aligned, banks spread
    constexpr int ROW = TD_HC_STEPS + 4;
    const int DH = p.dirs * p.H;
    const float *film = p.film + (size_t)(p.film_row ? p.film_row[0] : 0) * 2 * DH;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, hf = lane >> 5;
    // per channel: sum_f w'[c][f] (of the hi + lo values the layer kernel multiplied) and the constant term
    for (int c = wave; c < p.c_out; c += 4) {
        float ws = 0.f, cs = 0.f;
        for (int f = lane; f < DH; f += 64) {
            const float sc = film[f] + 1.0f, fw = p.fc_w[(size_t)c * DH + f];
            const float w = fw * p.ln_w[f] * sc;
            const __bf16 hi = (__bf16)w;
            ws += (float)hi + (float)(__bf16)(w - (float)hi);
            cs += fw * (p.ln_b[f] * sc + film[DH + f]);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { ws += __shfl_xor(ws, o); cs += __shfl_xor(cs, o); }
        if (lane == 0) { s_wsum[c] = ws; s_cst[c] = cs + p.fc_b[c]; }
    }
    __syncthreads();
    const int nreg = p.hp_stride > 12 ? 16 : 8;
    const int nchunk = td_cdiv_dev(p.L, TD_HC_STEPS);
    for (int work = blockIdx.x; work < p.nbb * nchunk; work += gridDim.x) {
        const int bblk = work % p.nbb, l0 = (work / p.nbb) * TD_HC_STEPS;
        const int nl = min(TD_HC_STEPS, p.L - l0);
        for (int li = wave; li < nl; li += 4) {
            const size_t task = (size_t)(l0 + li) * p.nbb + bblk;
            float acc[16], s1 = 0.f, s2 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            for (int d = 0; d < p.dirs; ++d) {
                const float *hp = p.hpart + ((task * p.dirs + d) * 64 + lane) * p.hp_stride;
                const f32x4 a0 = *(const f32x4 *)hp, a1 = *(const f32x4 *)(hp + 4), st = *(const f32x4 *)(hp + p.hp_stride - 4);
#pragma unroll
                for (int i = 0; i < 4; ++i) { acc[i] += a0[i]; acc[4 + i] += a1[i]; }
                if (nreg > 8) {
                    const f32x4 a2 = *(const f32x4 *)(hp + 8), a3 = *(const f32x4 *)(hp + 12);
#pragma unroll
                    for (int i = 0; i < 4; ++i) { acc[8 + i] += a2[i]; acc[12 + i] += a3[i]; }
                }
                s1 += st[0];
                s2 += st[1];
            }
            s1 += __shfl_xor(s1, 32);      // lanes m and m + 32 hold the two halves of sequence m's units
            s2 += __shfl_xor(s2, 32);
            const float mu = s1 / (float)DH;
            const float var = fmaxf(s2 / (float)DH - mu * mu, 0.f);
            const float rstd = rsqrtf(var + 1e-5f);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int c = (r & 3) + 8 * (r >> 2) + 4 * hf;     // accumulator row r of lane (m, hf)
                if (r < nreg && c < p.c_out) tile[(c * 32 + m) * ROW + li] = rstd * (acc[r] - mu * s_wsum[c]) + s_cst[c];
            }
        }
        __syncthreads();
        // (channel, sequence) rows of nl consecutive steps
        const bool vec = nl == TD_HC_STEPS && (p.L & 3) == 0;
        for (int i = threadIdx.x; i < p.c_out * 32 * (TD_HC_STEPS / 4); i += 256) {
            const int q = i % (TD_HC_STEPS / 4), seq = (i / (TD_HC_STEPS / 4)) % 32, c = i / (32 * (TD_HC_STEPS / 4));
            const int b = bblk * 32 + seq;
            if (b >= p.B) continue;
            const float *src = tile + (c * 32 + seq) * ROW + 4 * q;
            float *dst = p.y + ((size_t)b * p.c_out + c) * p.L + l0 + 4 * q;
            if (vec) *(f32x4 *)dst = *(const f32x4 *)src;
            else
                for (int k = 0; k < 4; ++k)
                    if (4 * q + k < nl) dst[k] = src[k];
        }
        __syncthreads();
    }
}

// parity/debug: fragments -> (B, L, D*H) fp32
__global__ void td_unpack_hidden_kernel(const char *act, float *hidden, int B, int L, int nbb, int H, int kh, int dirs) {
    const size_t total = (size_t)B * L * dirs * H;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int feat = (int)(idx % (dirs * H));
        const int l = (int)((idx / (dirs * H)) % L);
        const int b = (int)(idx / ((size_t)dirs * H * L));
        const int d = feat / H, unit = feat % H;
        const int ks = unit / 16, w = unit % 16;  // w = 8*(j>>2) + 4*hf + (j&3)
        const int hf = (w >> 2) & 1, j = ((w >> 3) << 2) | (w & 3);
        const int lane = hf * 32 + (b % 32);
        const __bf16 *fr = (const __bf16 *)(act + (((size_t)l * nbb + b / 32) * (dirs * kh) + d * kh + ks) * TD_FRAG_BYTES);
        hidden[idx] = (float)fr[lane * 8 + j];
    }
}

// ---------------------------------------------------------------------------------------------
// layer dispatch (one translation unit per kernel family: rnn_launch.h, rnn_layer_inst.hip)
// ---------------------------------------------------------------------------------------------
int td_launch_layer_infer(int gru, int nu, const TdLayerParams &p, bool layer0, int kxsel, hipStream_t stream, bool head = false) {
    return td_launch_layer_dispatch(gru, 0, nu, p, layer0, kxsel, stream, head);
}
int td_launch_layer_save(int gru, int nu, const TdLayerParams &p, bool layer0, int kxsel, hipStream_t stream);
// rnn_wide.hip: hidden sizes above 256
int td_launch_layer_wide(int gru, int save, int nu, const TdLayerParams &p, bool layer0, int kx0, hipStream_t stream);
int td_rnn_head_wide(const td_rnn_config *cfg, const TdRnnGeom &g, const float *const *misc, const char *act, const float *film,
                     int film_rows, const int32_t *film_row, int batch, int L, float *y, hipStream_t stream);

int td_profile_begin(int kind, void *stream);
void td_profile_end(int slot, void *stream);

// Runs pack-input + all recurrent layers.  acts[l] receives layer l's output fragments; saves[l] (training)
// the per-step gate activations.  Inference ping-pongs two activation buffers.
// fuse (optional): the top layer runs its fused-head variant on fuse->stream and leaves fuse->hpart instead of acts[top]
struct TdHeadFuseRun {
    const char *stream;
    size_t hdir_bytes;
    float *hpart;
    int hp_stride;
};
// [l_first, l_end): the layers to run (default: all); the layer-0 fragments are packed when l_first == 0
int td_rnn_run_layers_ex(const td_rnn_config *cfg, const void *packed, const float *x, int batch, int L, char *x0,
                         char *const *acts, float *const *saves, float *state, hipStream_t stream, const TdHeadFuseRun *fuse = nullptr,
                         int l_first = 0, int l_end = -1) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    const int nbb = td_cdiv(batch, TD_WAVE_BATCH);
    if (l_end < 0) l_end = cfg->layers;
    if (x && l_first == 0) {   // x == nullptr: x0 already holds the layer-0 fragments (the sampler's state path)
        size_t total = (size_t)L * nbb * g.kx0 * 64;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(td_pack_input_kernel, dim3(blocks), dim3(256), 0, stream, x, x0, batch, cfg->c_in, L, nbb, g.kx0);
        TD_LAUNCH_CHECK();
    }
    const char *in = l_first == 0 ? x0 : acts[l_first - 1];
    for (int l = l_first; l < l_end; ++l) {
        const TdLayerGeom &lg = l == 0 ? g.g0 : g.g1;
        TdLayerParams p;
        p.wpacked = (const char *)packed + g.layer_off[l];
        p.dir_stride = lg.dir_bytes;
        p.xin = in; p.out = acts[l];
        p.in_frags = lg.xfr; p.out_frags = g.dirs * g.kh;
        p.L = L; p.nbb = nbb; p.dirs = g.dirs;
        p.save = saves ? saves[l] : nullptr;
        p.save_dir_stride = (size_t)L * nbb * g.nu * 5 * 1024;
        p.state = state;
        p.state_dir_stride = td_state_dir_bytes(g, batch) / sizeof(float);
        p.hpart = nullptr; p.hp_stride = 0;
        const int kxsel = l == 0 ? g.kx0 : g.dirs;
        const bool head = fuse && l == cfg->layers - 1;
        if (head) {
            p.wpacked = fuse->stream; p.dir_stride = fuse->hdir_bytes;
            p.hpart = fuse->hpart; p.hp_stride = fuse->hp_stride;
        }
        if (td_layer_kernel_mode() == TD_LAYER_KERNEL_CLUSTER && !td_use_cluster(g, nbb, L)) {
            td_set_error("the cluster kernel was forced but cannot run hidden = %d, batch = %d, seq_len = %d", cfg->hidden, batch, L);
            return TD_ERR_UNSUPPORTED;
        }
        if (td_use_cluster(g, nbb, L)) {
            const int slot = l > 0 ? td_profile_begin(saves ? TD_PROFILE_LAYER_SAVE : TD_PROFILE_LAYER_INFER, stream) : -1;
            rc = td_launch_layer_cluster(cfg->gru, saves ? 1 : 0, g, p, (const char *)packed + g.cl_off[l], g.cl_dir[l == 0 ? 0 : 1],
                                         l == 0, (unsigned long long *)state, stream);
            td_profile_end(slot, stream);
        } else if (td_use_wide(g, nbb)) {
            const int slot = l > 0 ? td_profile_begin(saves ? TD_PROFILE_LAYER_SAVE : TD_PROFILE_LAYER_INFER, stream) : -1;
            rc = td_launch_layer_wide(cfg->gru, saves ? 1 : 0, g.nu, p, l == 0, g.kx0, stream);
            td_profile_end(slot, stream);
        } else if (saves) {
            const int slot = l > 0 ? td_profile_begin(TD_PROFILE_LAYER_SAVE, stream) : -1;
            rc = td_launch_layer_save(cfg->gru, g.nu, p, l == 0, kxsel, stream);
            td_profile_end(slot, stream);
        } else {
            const int slot = l > 0 ? td_profile_begin(TD_PROFILE_LAYER_INFER, stream) : -1;
            rc = td_launch_layer_infer(cfg->gru, g.nu, p, l == 0, kxsel, stream, head);
            td_profile_end(slot, stream);
        }
        if (rc != TD_OK) return rc;
        in = acts[l];
    }
    return TD_OK;
}

template <int KT>
static int td_launch_head(const TdHeadParams &p, hipStream_t stream) {
    const size_t ntask = (size_t)p.L * p.nbb;
    int blocks = (int)((ntask + 3) / 4);
    if (blocks > 2048) blocks = 2048;
    if (!p.film_row && p.film_rows > 1) {
        if (p.ffr) {
            hipLaunchKernelGGL(td_film_frag_kernel, dim3(td_cdiv(p.nbb * KT * 64, 256)), dim3(256), 0, stream, p.film, p.ln_w, p.ln_b,
                               (float *)p.ffr, p.B, p.nbb, p.H, p.kh, p.dirs);
            TD_LAUNCH_CHECK();
            hipLaunchKernelGGL((td_head_kernel<KT, 2>), dim3(8 * td_cdiv(blocks, 8)), dim3(256), 0, stream, p);
        } else {
            hipLaunchKernelGGL((td_head_kernel<KT, 1>), dim3(blocks), dim3(256), 0, stream, p);
        }
    } else {
        hipLaunchKernelGGL((td_head_kernel<KT, 0>), dim3(blocks), dim3(256), 0, stream, p);
    }
    TD_LAUNCH_CHECK();
    return TD_OK;
}

// film_frags: td_head_film_frag_bytes() of scratch for the fragment-ordered per-sequence FiLM (or NULL: the rows are read
// from `film` at every position)
size_t td_head_film_frag_bytes(const TdRnnGeom &g, int batch) {
    return g.dirs * g.kh > 32 ? 0 : (size_t)td_cdiv(batch, TD_WAVE_BATCH) * g.dirs * g.kh * 4096;
}

int td_rnn_head(const td_rnn_config *cfg, const float *const *misc, const char *act, const float *film, int film_rows,
                const int32_t *film_row, int batch, int L, float *y, hipStream_t stream, float *film_frags) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    if (g.dirs * g.kh > 32) return td_rnn_head_wide(cfg, g, misc, act, film, film_rows, film_row, batch, L, y, stream);
    TdHeadParams p;
    p.act = act;
    p.ln_w = misc[TD_LN_W]; p.ln_b = misc[TD_LN_B]; p.fc_w = misc[TD_FC_W]; p.fc_b = misc[TD_FC_B];
    p.film = film; p.film_row = film_row; p.film_rows = film_rows; p.ffr = film_frags;
    p.y = y; p.B = batch; p.L = L; p.nbb = td_cdiv(batch, TD_WAVE_BATCH);
    p.H = cfg->hidden; p.hpad = g.hpad; p.kh = g.kh; p.dirs = g.dirs; p.c_out = cfg->c_out;
    switch (g.dirs * g.kh) {
        case 2: return td_launch_head<2>(p, stream);
        case 4: return td_launch_head<4>(p, stream);
        case 6: return td_launch_head<6>(p, stream);
        case 8: return td_launch_head<8>(p, stream);
        case 12: return td_launch_head<12>(p, stream);
        case 16: return td_launch_head<16>(p, stream);
        case 24: return td_launch_head<24>(p, stream);
        case 32: return td_launch_head<32>(p, stream);
    }
    td_set_error("unsupported fragment count %d in the head", g.dirs * g.kh);
    return TD_ERR_UNSUPPORTED;
}

// Does a call take the fused head?  Chain: wherever the batch-block family runs the layers and the batch shares one FiLM
// row.  Tensor-level forward: only when td_set_head_fusion asked for it (td_rnn_read_hidden needs the activations).
static bool td_head_fusable(const td_rnn_config *cfg, const TdRnnGeom &g, int batch, int L, int film_rows, const int32_t *film_row,
                            bool chain) {
    const int mode = td_head_fusion_mode();
    if (mode == TD_HEAD_FUSION_OFF || (mode == TD_HEAD_FUSION_AUTO && !chain)) return false;
    const int nbb = td_cdiv(batch, TD_WAVE_BATCH);
    if (!(g.nu <= 8 && cfg->layers >= 2 && g.dirs * g.kh <= 32)) return false;
    if (!(film_row || film_rows == 1)) return false;
    return !td_use_cluster(g, nbb, L) && !td_use_wide(g, nbb);
}

// Sampler: the layer-0 fragments of the diffusion state (z, idx) into the denoiser workspace (they sit at its start);
// *kx0 = the k-tiles of the layer-0 input.  The chain calls it once, after the prior: from then on the reverse update's
// kernel leaves the next step's fragments there itself (diffusion.hip, FRAGS form).
int td_rnn_state_frags(const td_rnn_config *cfg, const td_diffusion_config *dcfg, const float *z, const uint8_t *idx, int batch,
                       int L, void *workspace, hipStream_t stream, int *kx0) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    int sumk = 0;
    for (int v = 0; v < dcfg->n_cat; ++v) sumk += dcfg->num_classes[v];
    TD_REQUIRE(dcfg->n_num + sumk == cfg->c_in && dcfg->seq_len == L, "diffusion state does not match the denoiser input");
    if (kx0) *kx0 = g.kx0;
    if (!z && !idx) return TD_OK;   // geometry only
    const int nbb = td_cdiv(batch, TD_WAVE_BATCH);
    const size_t sf_shm = (size_t)16 * g.kx0 * 32 * (TD_SF_LT + 1) * sizeof(float);
    if (sf_shm > 65536)
        TD_HIP(hipFuncSetAttribute((const void *)td_state_to_frags_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sf_shm));
    hipLaunchKernelGGL(td_state_to_frags_kernel, dim3(td_cdiv(L, TD_SF_LT), nbb), dim3(256), sf_shm, stream, *dcfg, sumk, z, idx,
                       (char *)workspace, batch, nbb, g.kx0);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

// The whole denoiser: layer-0 fragments (from x, from the sampler's diffusion state z / idx, or already in the workspace
// when x, z and idx are all NULL), the recurrent layers, the head -> y (B, C_out, L).
int td_rnn_denoise(const td_rnn_config *cfg, const void *packed, const float *const *misc, const float *x,
                   const td_diffusion_config *dcfg, const float *z, const uint8_t *idx, const float *film, int film_rows,
                   const int32_t *film_row, int batch, int L, float *y, void *workspace, hipStream_t stream, bool chain) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    size_t x0_bytes, act_bytes;
    td_ws_layout(g, batch, L, &x0_bytes, &act_bytes);
    char *x0 = (char *)workspace, *actA = x0 + x0_bytes, *actB = actA + act_bytes;
    const int nbb = td_cdiv(batch, TD_WAVE_BATCH);
    if (!x && (z || idx)) {   // sampler path: the layer-0 fragments come straight from the diffusion state
        rc = td_rnn_state_frags(cfg, dcfg, z, idx, batch, L, workspace, stream, nullptr);
        if (rc != TD_OK) return rc;
    }
    char *acts[16];
    for (int l = 0; l < cfg->layers; ++l) acts[l] = (l & 1) ? actB : actA;
    float *state = (float *)(actB + act_bytes);
    if (!td_head_fusable(cfg, g, batch, L, film_rows, film_row, chain)) {
        rc = td_rnn_run_layers_ex(cfg, packed, x, batch, L, x0, acts, nullptr, state, stream);
        if (rc != TD_OK) return rc;
        return td_rnn_head(cfg, misc, acts[cfg->layers - 1], film, film_rows, film_row, batch, L, y, stream, nullptr);
    }
    const TdHeadFuse hl = td_head_fuse_layout(cfg, g, batch, L);
    const int top = cfg->layers - 1;
    TdHeadFoldParams fp;
    fp.src = (const char *)packed + g.layer_off[top]; fp.src_dir = g.g1.dir_bytes;
    fp.dst = (char *)workspace + hl.stream_off; fp.dst_dir = hl.hdir_bytes;
    fp.ln_w = misc[TD_LN_W]; fp.fc_w = misc[TD_FC_W]; fp.film = film; fp.film_row = film_row;
    fp.nfb = g.g1.nf; fp.H = cfg->hidden; fp.kh = g.kh; fp.dirs = g.dirs; fp.c_out = cfg->c_out;
    hipLaunchKernelGGL(td_head_fold_kernel, dim3(256), dim3(256), 0, stream, fp);
    TD_LAUNCH_CHECK();
    TdHeadFuseRun run;
    run.stream = fp.dst; run.hdir_bytes = hl.hdir_bytes;
    run.hpart = (float *)((char *)workspace + hl.part_off); run.hp_stride = hl.hp_stride;
    rc = td_rnn_run_layers_ex(cfg, packed, x, batch, L, x0, acts, nullptr, state, stream, &run);
    if (rc != TD_OK) return rc;
    TdHeadCombineParams cp;
    cp.hpart = run.hpart; cp.ln_w = misc[TD_LN_W]; cp.ln_b = misc[TD_LN_B]; cp.fc_w = misc[TD_FC_W]; cp.fc_b = misc[TD_FC_B];
    cp.film = film; cp.film_row = film_row; cp.y = y; cp.hp_stride = hl.hp_stride;
    cp.B = batch; cp.L = L; cp.nbb = nbb; cp.H = cfg->hidden; cp.dirs = g.dirs; cp.c_out = cfg->c_out;
    const size_t nwork = (size_t)nbb * td_cdiv(L, TD_HC_STEPS);
    const size_t shm = (size_t)cfg->c_out * 32 * (TD_HC_STEPS + 4) * sizeof(float);
    TD_HIP(hipFuncSetAttribute((const void *)td_head_combine_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL(td_head_combine_kernel, dim3((unsigned)(nwork < 8192 ? nwork : 8192)), dim3(256), shm, stream, cp);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

extern "C" int td_rnn_forward(const td_rnn_config *cfg, const void *packed, const float *const *misc, const float *x,
                              const float *film, int32_t film_rows, const int32_t *film_row, int32_t batch,
                              int32_t seq_len, float *y, void *workspace, void *stream) {
    TD_REQUIRE(cfg && packed && misc && x && film && y && workspace, "NULL argument");
    TD_REQUIRE(batch > 0 && seq_len > 0, "batch and seq_len must be positive");
    TD_REQUIRE(film_rows == 1 || film_rows == batch || film_row, "film_rows must be 1 or batch");
    return td_rnn_denoise(cfg, packed, misc, x, nullptr, nullptr, nullptr, film, film_rows, film_row, batch, seq_len, y, workspace,
                          (hipStream_t)stream, false);
}

extern "C" int td_rnn_read_hidden(const td_rnn_config *cfg, const void *workspace, int32_t batch, int32_t seq_len,
                                  float *hidden, void *stream) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(workspace && hidden && batch > 0 && seq_len > 0, "bad argument");
    size_t x0_bytes, act_bytes;
    td_ws_layout(g, batch, seq_len, &x0_bytes, &act_bytes);
    const char *act = (const char *)workspace + x0_bytes + ((cfg->layers - 1) % 2) * act_bytes;
    size_t total = (size_t)batch * seq_len * g.dirs * cfg->hidden;
    int blocks = (int)((total + 255) / 256);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(td_unpack_hidden_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, act, hidden, batch, seq_len,
                       td_cdiv(batch, TD_WAVE_BATCH), cfg->hidden, g.kh, g.dirs);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

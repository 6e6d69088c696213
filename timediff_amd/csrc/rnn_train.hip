// Denoiser training path: forward that keeps what the backward pass needs, and the backward pass
// (autograd of models/ETDiff/blocks.py:333-341 as run by accelerator.backward, et_diff.py:237).
//
// Backward decomposition:
//   head_bwd      d y -> d h_top (bf16 fragments), d film, d LayerNorm, d fc           (one pass, LDS atomics)
//   per layer, top down:
//     bptt_sweep  ONE launch for all steps: gate-gradient arithmetic (d a_t as MFMA fragments + plain rows) and
//                 d h_{t-1} = W_hh^T d a_t on MFMA, state in registers, d a_t exchanged through LDS
//                 (up to 8192 sequences at H in (192, 256]: td_bptt_cluster_kernel, rnn_cluster_bwd.hip, 8 workgroups per
//                 batch block and direction with W_hh^T resident in registers)
//     frag_gemm   all steps at once: d x = W_ih^T d a         (gradient for the layer below)
//     dw_gemm     d W = d a . [x ; h_prev]^T over all (sequence, step) columns   (MFMA, split-K, fp32 atomics)
//     bias_reduce d b = row sums of d a   (H > 256 only: the sweeps up to 256 accumulate them in registers)
#include "rnn_cluster.h"
#include "rnn_launch.h"
#include "rng.h"

int td_profile_begin(int kind, void *stream);   // api.cpp
void td_profile_end(int slot, void *stream);

int td_launch_layer_save(int gru, int nu, const TdLayerParams &p, bool layer0, int kxsel, hipStream_t stream) {
    return td_launch_layer_dispatch(gru, 1, nu, p, layer0, kxsel, stream);
}

size_t td_cluster_xch_dir_bytes(const TdRnnGeom &g, int nbb);   // rnn_cluster.hip
int td_side_gemm_mode(void);                                     // api.cpp
int td_layer_kernel_mode(void);   // api.cpp
struct TdHeadFuseRun;   // rnn.hip: the sampler's fused head, not used in training
int td_rnn_run_layers_ex(const td_rnn_config *cfg, const void *packed, const float *x, int batch, int L, char *x0,
                         char *const *acts, float *const *saves, float *state, hipStream_t stream, const TdHeadFuseRun *fuse,
                         int l_first, int l_end);
int td_rnn_head(const td_rnn_config *cfg, const float *const *misc, const char *act, const float *film, int film_rows,
                const int32_t *film_row, int batch, int L, float *y, hipStream_t stream, float *film_frags);
size_t td_head_film_frag_bytes(const TdRnnGeom &g, int batch);

static inline size_t al256(size_t x) { return (x + 255) & ~(size_t)255; }

// split-K partial results of the weight-gradient GEMMs (td_dw_gemm) live in the training workspace
#define TD_DW_PARTIAL_BYTES ((size_t)64 << 20)

// ---------------------------------------------------------------------------------------------
// inter-layer dropout (nn.LSTM / nn.GRU `dropout=`, blocks.py:268-270: every layer's output but the last one's is
// multiplied by a Bernoulli(1 - p) mask / (1 - p) on its way to the next layer, in training mode)
// ---------------------------------------------------------------------------------------------
static float g_dropout_p = 0.f;
static uint64_t g_dropout_seed = 0;
float td_dropout_p(void) { return g_dropout_p; }

// Process-wide (like the kernel-selection knobs): read by td_rnn_forward_train, td_rnn_backward[_cb] and
// td_rnn_train_workspace_bytes; a backward pass must see the values its forward pass saw.
extern "C" int td_rnn_set_dropout(float p, uint64_t seed) {
    TD_REQUIRE(p >= 0.f && p < 1.f, "dropout probability must be in [0, 1), got %g", (double)p);
    g_dropout_p = p;
    g_dropout_seed = seed;
    return TD_OK;
}

// One thread per 16-byte fragment row (8 elements): ONE Philox block gives eight 16-bit uniforms; element j is kept where
// its field >= thresh = round(p * 65536).  The counter is (row index, layer), so the backward pass regenerates the mask of
// the forward pass on the gradient fragments (same layout) instead of reading a stored one.
__device__ __forceinline__ unsigned td_dropout_field(const Philox4 &r, int j) { return (r.v[j >> 1] >> (16 * (j & 1))) & 0xFFFFu; }

__global__ __launch_bounds__(256) void td_dropout_frags_kernel(bf16x8 *__restrict__ frags, size_t nvec, int layer, unsigned thresh,
                                                               float scale, uint64_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const Philox4 r = td_philox((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)layer, 0xD0D0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        bf16x8 v = frags[i];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = td_dropout_field(r, j) >= thresh ? (__bf16)((float)v[j] * scale) : (__bf16)0.f;
        frags[i] = v;
    }
}

// the mask itself, (B, L, D*H) fp32 with entries 0 or 1 / (1 - p): parity runs feed it to the oracle
__global__ __launch_bounds__(256) void td_dropout_mask_kernel(float *__restrict__ mask, int B, int L, int nbb, int H, int kh, int dirs,
                                                              int layer, unsigned thresh, float scale, uint64_t seed) {
    const size_t total = (size_t)B * L * dirs * H;
    for (size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (size_t)gridDim.x * 256) {
        const int feat = (int)(idx % (dirs * H)), l = (int)((idx / (dirs * H)) % L), b = (int)(idx / ((size_t)dirs * H * L));
        const int d = feat / H, unit = feat % H, ks = unit / 16, w = unit % 16, hf = (w >> 2) & 1, j = ((w >> 3) << 2) | (w & 3);
        const size_t i = (((size_t)l * nbb + b / 32) * (dirs * kh) + d * kh + ks) * 64 + hf * 32 + (b % 32);
        const Philox4 r = td_philox((uint32_t)i, (uint32_t)(i >> 32), (uint32_t)layer, 0xD0D0u, (uint32_t)seed, (uint32_t)(seed >> 32));
        mask[idx] = td_dropout_field(r, j) >= thresh ? scale : 0.f;
    }
}

static inline unsigned td_dropout_thresh(void) { return (unsigned)(g_dropout_p * 65536.0f + 0.5f); }
static inline float td_dropout_scale(void) { return 1.0f / (1.0f - (float)td_dropout_thresh() / 65536.0f); }

static int td_dropout_apply(char *frags, size_t bytes, int layer, hipStream_t stream) {
    const size_t nvec = bytes / 16;
    hipLaunchKernelGGL(td_dropout_frags_kernel, dim3((unsigned)((nvec + 255) / 256 < 8192 ? (nvec + 255) / 256 : 8192)), dim3(256), 0, stream,
                       (bf16x8 *)frags, nvec, layer, td_dropout_thresh(), td_dropout_scale(), g_dropout_seed);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

extern "C" int td_rnn_dropout_mask(const td_rnn_config *cfg, int32_t batch, int32_t seq_len, int32_t layer, float *mask, void *stream) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(mask && batch > 0 && seq_len > 0 && layer >= 0 && layer < cfg->layers - 1, "bad argument");
    hipLaunchKernelGGL(td_dropout_mask_kernel, dim3(4096), dim3(256), 0, (hipStream_t)stream, mask, batch, seq_len,
                       td_cdiv(batch, TD_WAVE_BATCH), cfg->hidden, g.kh, g.dirs, layer, td_dropout_thresh(), td_dropout_scale(),
                       g_dropout_seed);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

// ---------------------------------------------------------------------------------------------
// training workspace
// ---------------------------------------------------------------------------------------------
struct TdTrainLayout {
    int nbb, Bp, cpad;      // batch blocks, padded batch, layer-0 channels padded to 32
    size_t x0, act[16], save[16], plain[17], plainx[17], gout[16][2], daf[16][2], dapl[16][2], dc[2], inbox, chan, g1, dwpart, filmfr, ysave, pq, total;
    size_t act_bytes, save_dir_floats, daf_bytes, dapl_slab_elems, cols;
};

static TdTrainLayout td_train_layout(const td_rnn_config *cfg, const TdRnnGeom &g, int batch, int L) {
    TdTrainLayout t;
    t.nbb = td_cdiv(batch, TD_WAVE_BATCH);
    t.Bp = t.nbb * 32;
    t.cpad = 32 * td_cdiv(cfg->c_in, 32);
    t.cols = (size_t)L * t.Bp;
    t.act_bytes = (size_t)L * t.nbb * (g.dirs * g.kh) * TD_FRAG_BYTES;
    t.save_dir_floats = (size_t)L * t.nbb * g.nu * 5 * 1024;
    t.daf_bytes = (size_t)L * t.nbb * 4 * g.kh * TD_FRAG_BYTES;
    t.dapl_slab_elems = (size_t)g.hpad * t.cols;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off += al256(bytes); return o; };
    t.x0 = take((size_t)L * t.nbb * g.g0.xfr * TD_FRAG_BYTES);
    for (int l = 0; l < cfg->layers; ++l) t.act[l] = take(t.act_bytes);
    for (int l = 0; l < cfg->layers; ++l) t.save[l] = take(t.save_dir_floats * g.dirs * sizeof(float));
    t.plain[0] = take((size_t)t.cpad * t.cols * 2);
    for (int l = 1; l <= cfg->layers; ++l) t.plain[l] = take((size_t)g.dirs * g.hpad * t.cols * 2);
    // inter-layer dropout: layer l's input is the dropped copy of layer l-1's output (the weight gradients of W_ih read it),
    // while W_hh's read the output itself; without dropout the two are one buffer
    for (int l = 0; l <= cfg->layers; ++l)
        t.plainx[l] = (td_dropout_p() > 0.f && l >= 1 && l < cfg->layers) ? take((size_t)g.dirs * g.hpad * t.cols * 2) : t.plain[l];
    for (int l = 0; l < cfg->layers; ++l)
        for (int s = 0; s < 2; ++s) t.gout[l][s] = take(t.act_bytes);
    for (int l = 0; l < cfg->layers; ++l)
        for (int d = 0; d < g.dirs; ++d) {
            t.daf[l][d] = take(t.daf_bytes);
            t.dapl[l][d] = take(4 * t.dapl_slab_elems * 2);
        }
    {   // back to back (a multiple of 256 B each); the forward pass uses the pair as the layer kernels' state / exchange scratch
        size_t dcb = (size_t)t.nbb * g.nu * 1024 * sizeof(float);
        if (g.nu == TD_CLUSTER_NU && td_cluster_xch_dir_bytes(g, t.nbb) > dcb) dcb = td_cluster_xch_dir_bytes(g, t.nbb);
        for (int d = 0; d < 2; ++d) t.dc[d] = take(dcb);
    }
    t.chan = take((size_t)L * t.nbb * (cfg->c_out + 1) * 32 * sizeof(float));   // head backward: g * rstd per channel, mu
    t.g1 = take((size_t)batch * cfg->c_out * sizeof(float));
    t.inbox = take(td_bptt_cluster_inbox_bytes(g, t.nbb));   // the cluster BPTT sweep's partial-sum exchange
    t.dwpart = take(t.cols >= 4 * 512 ? TD_DW_PARTIAL_BYTES : 0);   // split-K partial results of the weight-gradient GEMMs
    t.filmfr = take(td_head_film_frag_bytes(g, batch));   // per-sequence FiLM in fragment order (head forward / backward)
    // the MFMA head backward reads the forward's output and two per-sequence tables (td_head_bwd_dh_mfma_kernel)
    t.ysave = take(td_head_film_frag_bytes(g, batch) ? (size_t)batch * cfg->c_out * L * sizeof(float) : 0);
    t.pq = take(td_head_film_frag_bytes(g, batch) ? (size_t)batch * cfg->c_out * 2 * sizeof(float) : 0);
    t.total = off;
    return t;
}

extern "C" size_t td_rnn_train_workspace_bytes(const td_rnn_config *cfg, int32_t batch, int32_t seq_len) {
    TdRnnGeom g;
    if (td_rnn_geom(cfg, &g) != TD_OK || batch <= 0 || seq_len <= 0) return 0;
    return td_train_layout(cfg, g, batch, seq_len).total;
}

// ---------------------------------------------------------------------------------------------
// plain (row-major, column = l*Bp + b) bf16 copies of the activations: operands of the dW GEMM
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void td_x_to_plain_kernel(const float *__restrict__ x, __bf16 *__restrict__ out, int B, int C,
                                                            int cpad, int L, int Bp) {
    const size_t total = (size_t)cpad * L * Bp;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int b = (int)(i % Bp), l = (int)((i / Bp) % L), c = (int)(i / ((size_t)Bp * L));
        float v = 0.f;
        if (b < B && c < C) v = x[((size_t)b * C + c) * L + l];
        out[i] = (__bf16)v;
    }
}

// act fragments [l][bblk][D*KH][64][8] -> rows (d*hpad + unit), columns l*Bp + b.  A workgroup takes one step, two
// neighbouring batch blocks and 16 fragments (256 units): 16-byte reads of whole fragments, transposed through LDS,
// 16-byte writes of whole 128-byte row segments (the first version gathered 2-byte elements: 0.6 TB/s).
#define TD_F2P_FRAGS 16
#define TD_F2P_ROW 72        // bf16 per LDS row: 64 sequences + 8 of padding (rows stay 16-byte aligned, banks spread)
__global__ __launch_bounds__(256) void td_frags_to_plain_kernel(const __bf16 *__restrict__ act, __bf16 *__restrict__ out, int L,
                                                                int nbb, int kh, int dirs) {
    __shared__ __attribute__((aligned(16))) __bf16 tile[TD_F2P_FRAGS * 16 * TD_F2P_ROW];
    const int KT = dirs * kh, Bp = nbb * 32, hpad = kh * 16;
    const size_t cols = (size_t)L * Bp;
    const int bb0 = blockIdx.x * 2, l = blockIdx.y, f0 = blockIdx.z * TD_F2P_FRAGS;
    const int nf = min(TD_F2P_FRAGS, KT - f0), nb = min(2, nbb - bb0);
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, m = lane & 31, hf = lane >> 5;
    for (int fi = wave; fi < 2 * TD_F2P_FRAGS; fi += 4) {
        const int blk = fi / TD_F2P_FRAGS, f = fi % TD_F2P_FRAGS;
        if (blk >= nb || f >= nf) continue;
        const bf16x8 v = *((const bf16x8 *)(act + (((size_t)l * nbb + bb0 + blk) * KT + f0 + f) * 512) + lane);
#pragma unroll
        for (int j = 0; j < 8; ++j) tile[(16 * f + 8 * (j >> 2) + 4 * hf + (j & 3)) * TD_F2P_ROW + blk * 32 + m] = v[j];
    }
    __syncthreads();
    const int seg = t & 7;               // 8 sequences = 16 bytes of a row
    if (seg * 8 >= nb * 32) return;
    for (int row = t >> 3; row < nf * 16; row += 32) {
        const int fg = f0 + row / 16, d = fg / kh, unit = 16 * (fg % kh) + row % 16;
        const bf16x8 v = *(const bf16x8 *)(tile + row * TD_F2P_ROW + seg * 8);
        *(bf16x8 *)(out + ((size_t)d * hpad + unit) * cols + (size_t)l * Bp + bb0 * 32 + seg * 8) = v;
    }
}

// ---------------------------------------------------------------------------------------------
// forward with saves
// ---------------------------------------------------------------------------------------------
extern "C" int td_rnn_forward_train(const td_rnn_config *cfg, const void *packed, const float *const *misc, const float *x,
                                    const float *film, int32_t batch, int32_t seq_len, float *y, void *workspace,
                                    void *stream_) {
    TD_REQUIRE(cfg && packed && misc && x && film && y && workspace, "NULL argument");
    TD_REQUIRE(batch > 0 && seq_len > 0, "batch and seq_len must be positive");
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const TdTrainLayout t = td_train_layout(cfg, g, batch, seq_len);
    char *ws = (char *)workspace;
    char *acts[16];
    float *saves[16];
    for (int l = 0; l < cfg->layers; ++l) {
        acts[l] = ws + t.act[l];
        saves[l] = (float *)(ws + t.save[l]);
    }
    // wide layers: the recurrent-state scratch is the d c carry region of the backward pass ([bblk][u][1024] fp32 per
    // direction, contiguous: td_train_layout).  Layer by layer: the plain (row-major) bf16 copies the weight-gradient GEMMs
    // read are taken right behind each layer, and with dropout the layer's output is then masked in place for the next one.
    const bool drop = td_dropout_p() > 0.f;
    for (int l = 0; l < cfg->layers; ++l) {
        rc = td_rnn_run_layers_ex(cfg, packed, x, batch, seq_len, ws + t.x0, acts, saves, (float *)(ws + t.dc[0]), stream, nullptr, l, l + 1);
        if (rc != TD_OK) return rc;
        const dim3 grid(td_cdiv(t.nbb, 2), seq_len, td_cdiv(g.dirs * g.kh, TD_F2P_FRAGS));
        hipLaunchKernelGGL(td_frags_to_plain_kernel, grid, dim3(256), 0, stream, (const __bf16 *)acts[l], (__bf16 *)(ws + t.plain[l + 1]),
                           seq_len, t.nbb, g.kh, g.dirs);
        TD_LAUNCH_CHECK();
        if (drop && l + 1 < cfg->layers) {
            rc = td_dropout_apply(acts[l], t.act_bytes, l, stream);
            if (rc != TD_OK) return rc;
            hipLaunchKernelGGL(td_frags_to_plain_kernel, grid, dim3(256), 0, stream, (const __bf16 *)acts[l],
                               (__bf16 *)(ws + t.plainx[l + 1]), seq_len, t.nbb, g.kh, g.dirs);
            TD_LAUNCH_CHECK();
        }
    }
    rc = td_rnn_head(cfg, misc, acts[cfg->layers - 1], film, batch, nullptr, batch, seq_len, y, stream,
                     td_head_film_frag_bytes(g, batch) ? (float *)(ws + t.filmfr) : nullptr);
    if (rc != TD_OK) return rc;
    if (td_head_film_frag_bytes(g, batch))
        TD_HIP(hipMemcpyAsync(ws + t.ysave, y, (size_t)batch * cfg->c_out * seq_len * sizeof(float), hipMemcpyDeviceToDevice, stream));
    hipLaunchKernelGGL(td_x_to_plain_kernel, dim3(2048), dim3(256), 0, stream, x, (__bf16 *)(ws + t.plain[0]), batch, cfg->c_in,
                       t.cpad, seq_len, t.Bp);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

// ---------------------------------------------------------------------------------------------
// transposed weight packing for the backward fragment GEMMs
//   WT[rb][term][ks][lane (r, hf)][j] = W[gate(term)*H + unit(ks, hf, j)][col(32 rb + r)]
// k (the contraction index) runs over gate units in the order the d a fragments hold them.
// ---------------------------------------------------------------------------------------------
struct TdPackTParams {
    const float *w;
    __bf16 *dst;
    int H, ld, kh, nterm, nrb, nu, hh;  // hh: columns are this layer's own hidden units; else [dir][unit] of the layer below
    int gate[4];
};

#define TD_PACKT_JOBS 12
struct TdPackTJobs {
    TdPackTParams job[TD_PACKT_JOBS];
};

__global__ __launch_bounds__(256) void td_pack_weights_T_kernel(TdPackTJobs jobs) {   // blockIdx.y: the job
    const TdPackTParams &p = jobs.job[blockIdx.y];
    const size_t total = (size_t)p.nrb * p.nterm * p.kh * 512;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int j = (int)(i % 8), lane = (int)((i / 8) % 64);
        const int ks = (int)((i / 512) % p.kh), term = (int)((i / (512 * (size_t)p.kh)) % p.nterm);
        const int rb = (int)(i / (512 * (size_t)p.kh * p.nterm));
        const int r = lane & 31, hf = lane >> 5;
        const int unit = 16 * ks + 8 * (j >> 2) + 4 * hf + (j & 3);
        int col = -1;
        if (p.hh) {
            if (32 * rb + r < p.H) col = 32 * rb + r;
        } else {
            const int dsrc = rb / p.nu, us = 32 * (rb % p.nu) + r;
            if (us < p.H) col = dsrc * p.H + us;
        }
        float v = 0.f;
        if (unit < p.H && col >= 0) v = p.w[(size_t)(p.gate[term] * p.H + unit) * p.ld + col];
        p.dst[i] = (__bf16)v;
    }
}

struct TdPackTGeom {
    size_t hh_bytes, ih_bytes, dir_bytes, layer_off[16], total;
    int ng;
};

static TdPackTGeom td_packT_geom(const td_rnn_config *cfg, const TdRnnGeom &g) {
    TdPackTGeom t;
    t.ng = cfg->gru ? 3 : 4;
    t.hh_bytes = (size_t)g.nu * t.ng * g.kh * TD_FRAG_BYTES;
    t.ih_bytes = (size_t)(g.dirs * g.nu) * t.ng * g.kh * TD_FRAG_BYTES;  // layers above the first only
    size_t off = 0;
    for (int l = 0; l < cfg->layers; ++l) {
        t.layer_off[l] = off;
        off += (t.hh_bytes + (l > 0 ? t.ih_bytes : 0)) * g.dirs;
    }
    t.total = off;
    return t;
}

extern "C" size_t td_rnn_packT_bytes(const td_rnn_config *cfg) {
    TdRnnGeom g;
    if (td_rnn_geom(cfg, &g) != TD_OK) return 0;
    return td_packT_geom(cfg, g).total;
}

extern "C" int td_rnn_pack_weights_T(const td_rnn_config *cfg, const float *const *lw, void *packT, void *stream) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(lw && packT, "NULL argument");
    const TdPackTGeom t = td_packT_geom(cfg, g);
    TdPackTJobs jobs;
    int nj = 0;
    auto push = [&](const TdPackTParams &p, bool last) {
        jobs.job[nj++] = p;
        if (nj == TD_PACKT_JOBS || last) {
            hipLaunchKernelGGL(td_pack_weights_T_kernel, dim3(1024, nj), dim3(256), 0, (hipStream_t)stream, jobs);
            nj = 0;
        }
    };
    for (int l = 0; l < cfg->layers; ++l)
        for (int d = 0; d < cfg->dirs; ++d) {
            const float *const *w = lw + (size_t)(l * cfg->dirs + d) * 4;
            char *base = (char *)packT + t.layer_off[l] + (size_t)d * (t.hh_bytes + (l > 0 ? t.ih_bytes : 0));
            const bool last = l == cfg->layers - 1 && d == cfg->dirs - 1;
            TdPackTParams p;
            p.H = cfg->hidden; p.kh = g.kh; p.nterm = t.ng; p.nu = g.nu;
            for (int i = 0; i < 4; ++i) p.gate[i] = i;
            // W_hh^T: rows of the result = own hidden units
            p.w = w[1]; p.ld = cfg->hidden; p.dst = (__bf16 *)base; p.nrb = g.nu; p.hh = 1;
            push(p, last && l == 0);
            TD_LAUNCH_CHECK();
            if (l > 0) {
                p.w = w[0]; p.ld = cfg->dirs * cfg->hidden; p.dst = (__bf16 *)(base + t.hh_bytes); p.nrb = g.dirs * g.nu; p.hh = 0;
                push(p, last);
                TD_LAUNCH_CHECK();
            }
        }
    return TD_OK;
}

// ---------------------------------------------------------------------------------------------
// head backward
// ---------------------------------------------------------------------------------------------
struct TdHeadBwdParams {
    const char *act;
    const float *ln_w, *ln_b, *fc_w, *film, *gy;
    char *gout;                                   // d h_top fragments (same layout as act)
    float *chan;                                  // [l][bblk][c_out + 1][32]: g[c] * rstd per output channel, then mu
    float *g1;                                    // [B][c_out]: sum over l of grad_y
    float *d_film, *d_ln_w, *d_ln_b, *d_fc_w, *d_fc_b;  // accumulated (atomics)
    int B, L, nbb, H, hpad, kh, dirs, c_out;
    int lch, llen;                                // parameter kernel: the steps are cut into lch chunks of llen
    const float *ffr, *ysave, *fc_b;              // MFMA variant: FiLM fragments (td_film_frag_kernel), the forward's y, fc bias
    float *pq;                                    // [B][c_out][2]: sum_f W[c][f] a_b[f], sum_f W[c][f] b_b[f]
};

// sum over the 32 lanes that share bit 5 of the lane id (one half-wave = the 32 sequences of a batch block)
__device__ __forceinline__ float td_half_wave_sum(float x) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) x += __shfl_xor(x, o);
    return x;
}

// The head is y = fc(FiLM(LayerNorm(h))) per position (blocks.py:336-341).  Its backward pass in three kernels, none of
// which reduces anything per position across lanes or through atomics (the first version did both: 570 M global
// atomics at batch 2048, 14 ms):
//   td_head_bwd_dh_kernel     per position: LayerNorm statistics, d h (fragments), and g[c] * rstd, mu laid out per
//                             (step, batch block, channel) for the kernel below
//   td_rowsum_kernel          G1[b][c] = sum_l grad_y[b][c][l]
//   td_head_bwd_param_kernel  a wave owns (batch block, 16-unit fragment) and walks the steps, accumulating per lane
//                                 X[c][f] = sum_l g[c] * rstd * (h[f] - mu)     ( = sum_l g[c] * xhat[f] )
//                             in registers; every parameter gradient is a combination of X and G1 (derivation in
//                             DESIGN.md section 4.4), reduced over the 32 sequences ONCE, at the end.

// ROG: LayerNorm weight and fc_w are read from global memory / L2 instead of LDS (wide hidden sizes with many channels)
template <int CMAX, bool ROG>
__global__ __launch_bounds__(256) void td_head_bwd_dh_kernel(TdHeadBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) float hl[];
    const int DHP = p.dirs * p.hpad, DH = p.dirs * p.H;
    float *s_lnw = hl, *s_fcw = hl + DHP;
    if constexpr (!ROG) {
        for (int i = threadIdx.x; i < DHP; i += 256) {
            const int d = i / p.hpad, unit = i % p.hpad;
            const bool ok = unit < p.H;
            s_lnw[i] = ok ? p.ln_w[d * p.H + unit] : 0.f;
            for (int c = 0; c < CMAX; ++c) s_fcw[c * DHP + i] = (ok && c < p.c_out) ? p.fc_w[(size_t)c * DH + d * p.H + unit] : 0.f;
        }
        __syncthreads();
    }
    auto lnw = [&](int ip, int feat) { return ROG ? p.ln_w[feat] : s_lnw[ip]; };
    auto fcw = [&](int c, int ip, int feat) { return ROG ? (c < p.c_out ? p.fc_w[(size_t)c * DH + feat] : 0.f) : s_fcw[c * DHP + ip]; };
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, hf = lane >> 5;
    const int KT = p.dirs * p.kh, CH = p.c_out + 1;
    const size_t ntask = (size_t)p.L * p.nbb;
    for (size_t task = (size_t)blockIdx.x * 4 + wave; task < ntask; task += (size_t)gridDim.x * 4) {
        const int l = (int)(task / p.nbb), bblk = (int)(task % p.nbb);
        const int b = bblk * 32 + m;
        const bool valid = b < p.B;
        const bf16x8 *fp = (const bf16x8 *)(p.act + task * KT * TD_FRAG_BYTES) + lane;
        float sum = 0.f;
        for (int ks = 0; ks < KT; ++ks) {
            const bf16x8 v = fp[ks * 64];
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += (float)v[j];
        }
        sum += __shfl_xor(sum, 32);
        const float mu = sum / (float)DH;
        float sq = 0.f;
        for (int ks = 0; ks < KT; ++ks) {
            const bf16x8 v = fp[ks * 64];
            const int unit0 = 16 * (ks % p.kh) + 4 * hf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dlt = (float)v[j] - mu;
                sq += (unit0 + 8 * (j >> 2) + (j & 3)) < p.H ? dlt * dlt : 0.f;
            }
        }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq / (float)DH + 1e-5f);
        float g[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) g[c] = (valid && c < p.c_out) ? p.gy[((size_t)b * p.c_out + c) * p.L + l] : 0.f;
        if (hf == 0) {
            float *cp = p.chan + task * CH * 32 + m;
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < p.c_out) cp[c * 32] = g[c] * rstd;
            cp[p.c_out * 32] = mu;
        }
        const float *film = p.film + (size_t)(valid ? b : 0) * 2 * DH;
        // the two means of the LayerNorm backward
        float m1 = 0.f, m2 = 0.f;
        for (int ks = 0; ks < KT; ++ks) {
            const bf16x8 v = fp[ks * 64];
            const int d = ks / p.kh;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int unit = 16 * (ks % p.kh) + 8 * (j >> 2) + 4 * hf + (j & 3);
                if (unit >= p.H) continue;
                const int ip = d * p.hpad + unit, feat = d * p.H + unit;
                float dyf = 0.f;
#pragma unroll
                for (int c = 0; c < CMAX; ++c) dyf += g[c] * fcw(c, ip, feat);
                const float dxh = dyf * (film[feat] + 1.0f) * lnw(ip, feat);
                m1 += dxh;
                m2 += dxh * (((float)v[j] - mu) * rstd);
            }
        }
        m1 += __shfl_xor(m1, 32);
        m2 += __shfl_xor(m2, 32);
        m1 /= (float)DH;
        m2 /= (float)DH;
        // d h
        bf16x8 *op = (bf16x8 *)(p.gout + task * KT * TD_FRAG_BYTES) + lane;
        for (int ks = 0; ks < KT; ++ks) {
            const bf16x8 v = fp[ks * 64];
            const int d = ks / p.kh;
            bf16x8 o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int unit = 16 * (ks % p.kh) + 8 * (j >> 2) + 4 * hf + (j & 3);
                float dh = 0.f;
                if (unit < p.H && valid) {
                    const int ip = d * p.hpad + unit, feat = d * p.H + unit;
                    const float xh = ((float)v[j] - mu) * rstd;
                    float dyf = 0.f;
#pragma unroll
                    for (int c = 0; c < CMAX; ++c) dyf += g[c] * fcw(c, ip, feat);
                    const float dxh = dyf * (film[feat] + 1.0f) * lnw(ip, feat);
                    dh = rstd * (dxh - m1 - xh * m2);
                }
                o[j] = (__bf16)dh;
            }
            op[ks * 64] = o;
        }
    }
}

// ---- d h through the matrix cores (hidden sizes whose position fits a wave's registers: D * KH <= 32 fragments) ----
// With the per-sequence multipliers a = ln_w (1 + scale_b), b = ln_b (1 + scale_b) + shift_b (fragment order, left by the
// forward pass: td_film_frag_kernel) the head is y[c] = sum_f W[c][f] (xhat[f] a[f] + b[f]) + fc_b[c], so
//     d xhat[f] = a[f] * dyf[f],   dyf = W^T g            one MFMA per 32 features: A = W^T fragments (LDS), B = g (hi + lo)
//     m1 = mean_f d xhat[f]         = sum_c g[c] P[b][c] / DH,              P[b][c] = sum_f W[c][f] a_b[f]
//     m2 = mean_f d xhat[f] xhat[f] = sum_c g[c] (y[c] - fc_b[c] - Q[b][c]) / DH,   Q[b][c] = sum_f W[c][f] b_b[f]
//     d h[f] = rstd (d xhat[f] - m1 - xhat[f] m2)
// i.e. both LayerNorm means come from per-position quantities the forward pass left (y) and two per-sequence tables, and
// the position's fragments are read once.  The accumulator of the MFMA holds dyf in exactly the order of the activation
// fragments (rows 0..15 = fragment 2 fb, rows 16..31 = fragment 2 fb + 1).
__global__ __launch_bounds__(256) void td_head_pq_kernel(TdHeadBwdParams p) {
    // one wave per (sequence, channel): the lanes split the D H features
    const int idx = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (idx >= p.B * p.c_out) return;
    const int b = idx / p.c_out, c = idx % p.c_out, KT = p.dirs * p.kh, DH = p.dirs * p.H;
    const int bblk = b / 32, m = b % 32;
    float P = 0.f, Q = 0.f;
    for (int feat = lane; feat < DH; feat += 64) {
        const int d = feat / p.H, unit = feat % p.H, ks = d * p.kh + unit / 16, w = unit % 16;
        const int hf = (w >> 2) & 1, j = ((w >> 3) << 2) | (w & 3);
        const float *f = p.ffr + (((size_t)bblk * KT + ks) * 2) * 512 + (hf * 32 + m) * 8 + j;
        const float wv = p.fc_w[(size_t)c * DH + feat];
        P = fmaf(wv, f[0], P);
        Q = fmaf(wv, f[512], Q);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { P += __shfl_xor(P, o); Q += __shfl_xor(Q, o); }
    if (lane == 0) {
        p.pq[(size_t)idx * 2] = P;
        p.pq[(size_t)idx * 2 + 1] = Q;
    }
}

template <int KT, int KC>
__global__ __launch_bounds__(256) void td_head_bwd_dh_mfma_kernel(TdHeadBwdParams p) {
    extern __shared__ __attribute__((aligned(16))) char wfr[];   // [KT / 2][KC][hi | lo] A fragments of W^T
    const int DH = p.dirs * p.H;
    // (consecutive threads read consecutive features of one channel's row: coalesced; the 2-byte LDS writes scatter instead)
    for (int idx = threadIdx.x; idx < (KT / 2) * KC * 64 * 8; idx += 256) {
        const int r = idx % 32, j = (idx / 32) % 8, hf = (idx / 256) % 2, kc = (idx / 512) % KC, fb = idx / (512 * KC);
        const int ln = hf * 32 + r, c = 16 * kc + 8 * hf + j;
        const int ks0 = 2 * fb, d = ks0 / p.kh, unit = 16 * (ks0 % p.kh) + r;
        const float w = (c < p.c_out && unit < p.H) ? p.fc_w[(size_t)c * DH + d * p.H + unit] : 0.f;
        const __bf16 hi = (__bf16)w;
        __bf16 *dst = (__bf16 *)(wfr + (size_t)((fb * KC + kc) * 2) * TD_FRAG_BYTES);
        dst[ln * 8 + j] = hi;
        dst[512 + ln * 8 + j] = (__bf16)(w - (float)hi);
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, hf = lane >> 5;
    const int CH = p.c_out + 1;
    // batch blocks dealt to the XCDs as in the forward head (td_head_kernel, MODE 2)
    const int xcd = blockIdx.x % 8, nbx = (p.nbb - xcd + 7) / 8;
    const size_t ntask = (size_t)p.L * nbx;
    for (size_t tx = (size_t)(blockIdx.x / 8) * 4 + wave; tx < ntask; tx += (size_t)(gridDim.x / 8) * 4) {
        const int l = (int)(tx / nbx), bblk = xcd + 8 * (int)(tx % nbx);
        const size_t task = (size_t)l * p.nbb + bblk;
        const int b = bblk * 32 + m;
        const bool valid = b < p.B;
        const bf16x8 *fp = (const bf16x8 *)(p.act + task * KT * TD_FRAG_BYTES) + lane;
        bf16x8 v[KT];
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) v[ks] = fp[ks * 64];
        // this lane's channels: c = 16 kc + 8 hf + j
        float g[KC][8];
        float m1 = 0.f, m2 = 0.f;
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int c = 16 * kc + 8 * hf + j;
                const bool ok = valid && c < p.c_out;
                const size_t at = ((size_t)(ok ? b : 0) * p.c_out + (ok ? c : 0)) * p.L + l;
                g[kc][j] = ok ? p.gy[at] : 0.f;
                if (ok) {
                    const float *pq = p.pq + ((size_t)b * p.c_out + c) * 2;
                    m1 = fmaf(g[kc][j], pq[0], m1);
                    m2 = fmaf(g[kc][j], p.ysave[at] - p.fc_b[c] - pq[1], m2);
                }
            }
        m1 += __shfl_xor(m1, 32);
        m2 += __shfl_xor(m2, 32);
        m1 /= (float)DH;
        m2 /= (float)DH;
        float sum = 0.f;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks)
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += (float)v[ks][j];
        sum += __shfl_xor(sum, 32);
        const float mu = sum / (float)DH;
        float sq = 0.f;
#pragma unroll
        for (int ks = 0; ks < KT; ++ks) {
            const int unit0 = 16 * (ks % p.kh) + 4 * hf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float dlt = (float)v[ks][j] - mu;
                sq += (unit0 + 8 * (j >> 2) + (j & 3)) < p.H ? dlt * dlt : 0.f;
            }
        }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq / (float)DH + 1e-5f);
        {   // what the parameter kernel reads: g[c] * rstd per channel, then mu
            float *cp = p.chan + task * CH * 32 + m;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int c = 16 * kc + 8 * hf + j;
                    if (c < p.c_out) cp[c * 32] = g[kc][j] * rstd;
                }
            if (hf == 0) cp[p.c_out * 32] = mu;
        }
        bf16x8 gh[KC], gl[KC];
#pragma unroll
        for (int kc = 0; kc < KC; ++kc)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const __bf16 h = (__bf16)g[kc][j];
                gh[kc][j] = h;
                gl[kc][j] = (__bf16)(g[kc][j] - (float)h);
            }
        bf16x8 *op = (bf16x8 *)(p.gout + task * KT * TD_FRAG_BYTES) + lane;
        const char *wf = wfr + lane * 16;
        // the FiLM multipliers of block fb + 1 are fetched while block fb is worked (the blocks are pinned in order, below)
        f32x4 am[2][2][2];
        auto load_a = [&](int fb, f32x4 (&dst)[2][2]) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const f32x4 *fa = (const f32x4 *)(p.ffr + (((size_t)bblk * KT + 2 * fb + half) * 2) * 512) + lane * 2;
                dst[half][0] = fa[0];
                dst[half][1] = fa[1];
            }
        };
        load_a(0, am[0]);
#pragma unroll
        for (int fb = 0; fb < KT / 2; ++fb) {
            if (fb + 1 < KT / 2) load_a(fb + 1, am[(fb + 1) & 1]);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int kc = 0; kc < KC; ++kc) {
                const bf16x8 wh = *(const bf16x8 *)(wf + (size_t)((fb * KC + kc) * 2) * TD_FRAG_BYTES);
                const bf16x8 wl = *(const bf16x8 *)(wf + (size_t)((fb * KC + kc) * 2 + 1) * TD_FRAG_BYTES);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl, gh[kc], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, gl[kc], acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh, gh[kc], acc, 0, 0, 0);
            }
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                const int ks = 2 * fb + half;
                const f32x4 a0 = am[fb & 1][half][0], a1 = am[fb & 1][half][1];
                bf16x8 o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int unit = 16 * (ks % p.kh) + 8 * (j >> 2) + 4 * hf + (j & 3);
                    const float xh = ((float)v[ks][j] - mu) * rstd;
                    const float dxh = acc[8 * half + j] * (j < 4 ? a0[j & 3] : a1[j & 3]);
                    o[j] = (__bf16)((valid && unit < p.H) ? rstd * (dxh - m1 - xh * m2) : 0.f);
                }
                op[ks * 64] = o;
            }
            __builtin_amdgcn_sched_barrier(0);   // keeps the FiLM loads of later blocks from being hoisted (register pressure)
        }
    }
}

template <int KT>
static int td_launch_head_bwd_mfma(const TdHeadBwdParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(td_head_pq_kernel, dim3(td_cdiv(p.B * p.c_out, 4)), dim3(256), 0, stream, p);
    TD_LAUNCH_CHECK();
    const size_t ntask = (size_t)p.L * p.nbb;
    int blocks = (int)((ntask + 3) / 4);
    if (blocks > 2048) blocks = 2048;
    blocks = 8 * td_cdiv(blocks, 8);
    if (p.c_out <= 16) {
        const size_t shm = (size_t)(KT / 2) * 2 * TD_FRAG_BYTES;
        TD_HIP(hipFuncSetAttribute((const void *)td_head_bwd_dh_mfma_kernel<KT, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL((td_head_bwd_dh_mfma_kernel<KT, 1>), dim3(blocks), dim3(256), shm, stream, p);
    } else {
        const size_t shm = (size_t)(KT / 2) * 4 * TD_FRAG_BYTES;
        TD_HIP(hipFuncSetAttribute((const void *)td_head_bwd_dh_mfma_kernel<KT, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL((td_head_bwd_dh_mfma_kernel<KT, 2>), dim3(blocks), dim3(256), shm, stream, p);
    }
    TD_LAUNCH_CHECK();
    return TD_OK;
}

static int td_head_bwd_mfma(const TdHeadBwdParams &p, hipStream_t stream) {
    switch (p.dirs * p.kh) {
        case 2: return td_launch_head_bwd_mfma<2>(p, stream);
        case 4: return td_launch_head_bwd_mfma<4>(p, stream);
        case 6: return td_launch_head_bwd_mfma<6>(p, stream);
        case 8: return td_launch_head_bwd_mfma<8>(p, stream);
        case 12: return td_launch_head_bwd_mfma<12>(p, stream);
        case 16: return td_launch_head_bwd_mfma<16>(p, stream);
        case 24: return td_launch_head_bwd_mfma<24>(p, stream);
        case 32: return td_launch_head_bwd_mfma<32>(p, stream);
    }
    td_set_error("head backward: %d fragments per position have no MFMA variant", p.dirs * p.kh);
    return TD_ERR_UNSUPPORTED;
}

// G1[row] = sum of the row's L entries (rows = B * c_out rows of grad_y), one wave per row
__global__ __launch_bounds__(256) void td_rowsum_kernel(const float *__restrict__ x, float *__restrict__ out, int rows, int L) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    float s = 0.f;
    for (int l = lane; l < L; l += 64) s += x[(size_t)row * L + l];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) out[row] = s;
}

// wave = (batch block blockIdx.x, fragment blockIdx.y * 4 + wave, channels [NC * (blockIdx.z / lch), + NC), steps of chunk
// blockIdx.z % lch): small batches are cut along the steps so that the launch still has a few thousand waves; every result
// is accumulated with atomics, and the terms in G1 (whole-sequence sums) are added by chunk 0 alone
template <int NC>
__global__ __launch_bounds__(256, 2) void td_head_bwd_param_kernel(TdHeadBwdParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, hf = lane >> 5;
    const int KT = p.dirs * p.kh, CH = p.c_out + 1, DH = p.dirs * p.H;
    const int bblk = blockIdx.x, ks = blockIdx.y * 4 + wave, c0 = (blockIdx.z / p.lch) * NC, lc = blockIdx.z % p.lch;
    if (ks >= KT) return;
    const int l0 = lc * p.llen, l1 = min(p.L, l0 + p.llen);
    const int b = bblk * 32 + m;
    const bool valid = b < p.B;
    float X[NC][8];
#pragma unroll
    for (int c = 0; c < NC; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) X[c][j] = 0.f;
    const bf16x8 *fp = (const bf16x8 *)(p.act + ((size_t)bblk * KT + ks) * TD_FRAG_BYTES) + lane;
    const float *cp = p.chan + (size_t)bblk * CH * 32 + m;
    const size_t fstep = (size_t)p.nbb * KT * 64, cstep = (size_t)p.nbb * CH * 32;
    int coff[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) coff[c] = (c0 + c < p.c_out ? c0 + c : p.c_out - 1) * 32;
#pragma unroll 2
    for (int l = l0; l < l1; ++l) {
        const bf16x8 v = fp[(size_t)l * fstep];
        const float *cl = cp + (size_t)l * cstep;
        const float mu = cl[p.c_out * 32];
        float gr[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) gr[c] = cl[coff[c]];   // rows beyond c_out repeat the last channel and are ignored below
        float hc[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) hc[j] = (float)v[j] - mu;
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) X[c][j] = __builtin_fmaf(gr[c], hc[j], X[c][j]);
    }
    float G1[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) G1[c] = (valid && lc == 0 && c0 + c < p.c_out) ? p.g1[(size_t)b * p.c_out + c0 + c] : 0.f;
    const float *film = p.film + (size_t)(valid ? b : 0) * 2 * DH;
    float *dfilm = p.d_film + (size_t)(valid ? b : 0) * 2 * DH;
    const int d = ks / p.kh;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int unit = 16 * (ks % p.kh) + 8 * (j >> 2) + 4 * hf + (j & 3);
        if (unit >= p.H) continue;     // uniform over the half-wave
        const int feat = d * p.H + unit;
        const float lw = p.ln_w[feat], lb = p.ln_b[feat];
        const float sc1 = film[feat] + 1.0f, sh = film[DH + feat];
        float sx = 0.f, sg = 0.f;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            if (c0 + c >= p.c_out) continue;
            const float fw = p.fc_w[(size_t)(c0 + c) * DH + feat];
            sx += fw * X[c][j];
            sg += fw * G1[c];
            // d fc_w[c][f] = sum over positions of g[c] * (lnout * (1 + scale) + shift)
            const float t = td_half_wave_sum(sc1 * (lw * X[c][j] + lb * G1[c]) + sh * G1[c]);
            if (m == 0) atomicAdd(&p.d_fc_w[(size_t)(c0 + c) * DH + feat], t);
        }
        if (valid) {
            atomicAdd(&dfilm[feat], lw * sx + lb * sg);     // d scale = sum_l dyf * lnout
            atomicAdd(&dfilm[DH + feat], sg);               // d shift = sum_l dyf
        }
        const float t_lnw = td_half_wave_sum(sc1 * sx), t_lnb = td_half_wave_sum(sc1 * sg);
        if (m == 0) {
            atomicAdd(&p.d_ln_w[feat], t_lnw);
            atomicAdd(&p.d_ln_b[feat], t_lnb);
        }
    }
    if (ks == 0 && lc == 0) {
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const float t = td_half_wave_sum(G1[c]);
            if (lane == 0 && c0 + c < p.c_out) atomicAdd(&p.d_fc_b[c0 + c], t);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// operands of a layer's BPTT sweep
// ---------------------------------------------------------------------------------------------
struct TdBpttParams {
    const char *gsrc[2];     // gradient w.r.t. this layer's output (fragments), n_src of them are summed
    int n_src;
    const float *save;       // [dir][l][bblk][NU][5][64][16]
    char *daf;               // [dir][l][bblk][4][KH] fragments
    __bf16 *dapl;            // [dir][4][hpad][L*Bp]
    size_t save_dir, daf_dir, dapl_dir;   // direction strides (floats / bytes / elems)
    float *db_ih[2], *db_hh[2];   // per direction: bias gradients (accumulated; the single-block-per-wave sweeps fold the row sums of d a in)
    int L, nbb, nu, kh, dirs, out_frags, gru, H;
};

// ---------------------------------------------------------------------------------------------
// fragment GEMM: out[tile][2 rb + {0,1}] = sum_{term, ks} WT[rb][term][ks] x in[tile][slab(term) * KH + ks]
// ---------------------------------------------------------------------------------------------
struct TdFragGemmParams {
    const char *wt;          // [rb][nterm][kh] fragments
    const char *in;          // [tile][in_frags] fragments
    char *out;               // [tile][out_frags] fragments; row block rb lands at frag out_off + 2 rb
    size_t in_tile_off;      // fragments to skip at the start (selects one time step of a [l][bblk] array)
    int in_frags, out_frags, out_off, ntiles, nrb, nterm, kh;
    int slab[4];
};

__global__ __launch_bounds__(256) void td_frag_gemm_kernel(TdFragGemmParams p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int rb = blockIdx.x * 4 + wave, tile = blockIdx.y;
    if (rb >= p.nrb) return;
    const bf16x8 *wp = (const bf16x8 *)(p.wt + (size_t)rb * p.nterm * p.kh * TD_FRAG_BYTES) + lane;
    const bf16x8 *ip = (const bf16x8 *)(p.in + (p.in_tile_off + (size_t)tile * p.in_frags) * TD_FRAG_BYTES) + lane;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int term = 0; term < p.nterm; ++term) {
        const bf16x8 *bp = ip + (size_t)p.slab[term] * p.kh * 64;
        const bf16x8 *ap = wp + (size_t)term * p.kh * 64;
#pragma unroll 4
        for (int ks = 0; ks < p.kh; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[ks * 64], bp[ks * 64], acc, 0, 0, 0);
    }
    bf16x8 a, b;
#pragma unroll
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)acc[j]; b[j] = (__bf16)acc[8 + j]; }
    bf16x8 *op = (bf16x8 *)(p.out + ((size_t)tile * p.out_frags + p.out_off + 2 * rb) * TD_FRAG_BYTES) + lane;
    op[0] = a;
    op[64] = b;
}

// The same product with the weights resident: wave w of a workgroup keeps the NT * KH fragments of row block 4 rbg + w in
// registers for all the tiles of its chunk, and the input fragments are staged in LDS once per workgroup by LDS-DMA: a
// ring of 10 stages of KH fragments (one gate slab of one tile, 16 KiB), filled 8 stages = two tiles ahead (HBM latency
// under load exceeds the time a tile's 64 MFMAs take), read 8 fragments ahead of the MFMAs.  Every input byte is fetched
// once per group of four row blocks instead of once per row block, and the weights once per workgroup instead of once
// per tile.  The kernel above re-reads both operands for every MFMA (2 KiB per MFMA and wave: L2-bound, 0.37 PFLOP/s).
#define TD_FG_STAGES 10
template <int N, class F>
__device__ __forceinline__ void td_unrolled(F &&f) {
    if constexpr (N > 0) {
        td_unrolled<N - 1>(f);
        f(std::integral_constant<int, N - 1>{});
    }
}

// KS = 2 (hidden sizes whose NT * KH fragments per row block exceed a wave's registers: KH = 32): two waves share a row
// block, each holding the weight fragments of one half of every gate slab's k-steps and reading only that half of a stage;
// the odd wave hands its accumulator to the even one through LDS at the end of a tile.  NSTG: ring stages (KH KiB each).
template <int NT, int KH, int KS, int NSTG>
__global__ __launch_bounds__(256, 1) void td_frag_gemm_res_kernel(TdFragGemmParams p, int tiles_per_wg) {
    constexpr int KHL = KH / KS, NK = NT * KHL, LEAD = NSTG - 2, RING = 8, PW = KH / 4, RBG = 4 / KS;
    static_assert(KH % 4 == 0 && KH % KS == 0 && RING <= KHL && NK % RING == 0 && (KS == 1 || KS == 2), "stage / ring geometry");
    extern __shared__ __attribute__((aligned(16))) char sm[];   // NSTG x KH KiB (+ 2 x 4 KiB of accumulators, KS = 2)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int kh_half = wave % KS;                      // which half of the k-steps this wave multiplies
    // workgroups go to the XCDs round-robin by index: the rbg workgroups that read the same chunk of tiles sit on ONE XCD
    // (indices 8 apart), so the chunk comes from HBM once and from that XCD's L2 for the others
    const int nrbg = (p.nrb + RBG - 1) / RBG;
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int rbg = slot % nrbg, chunk = (slot / nrbg) * 8 + xcd;
    const int rb0 = rbg * RBG + wave / KS;
    const int rb = rb0 < p.nrb ? rb0 : p.nrb - 1;      // a wave past the last row block repeats it (and stores nothing)
    const int t0 = chunk * tiles_per_wg, t1 = min(p.ntiles, t0 + tiles_per_wg);
    if (t0 >= t1) return;
    const int nst = (t1 - t0) * NT;                     // stages of this chunk
    bf16x8 A[NK];
    {
        const bf16x8 *wp = (const bf16x8 *)(p.wt + (size_t)rb * NT * KH * TD_FRAG_BYTES) + lane;
#pragma unroll
        for (int i = 0; i < NK; ++i) A[i] = wp[(size_t)((i / KHL) * KH + kh_half * KHL + i % KHL) * 64];
    }
    // the gate slab of term q, four bits each (an indexed local array would live in scratch memory)
    unsigned slabs = 0;
#pragma unroll
    for (int i = 0; i < NT; ++i) slabs |= (unsigned)p.slab[i] << (4 * i);
    const unsigned lane16 = lane * 16;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)sm;
    // stage j = gate slab j % NT of tile t0 + j / NT; wave w moves its pieces [w PW, (w + 1) PW).  M0 = LDS base of the
    // piece (used by nothing else here); issued from inline asm so that the compiler does not drain vmcnt / lgkmcnt at
    // the next LDS access (rnn_layer.h)
    auto stage_dma = [&](int j) {
        const int tile = t0 + j / NT;
        const int sl = (int)((slabs >> (4 * (j % NT))) & 15u);
        const char *src = p.in + (p.in_tile_off + (size_t)tile * p.in_frags + (size_t)sl * KH + wave * PW) * TD_FRAG_BYTES;
        const unsigned dst = lds0 + (unsigned)(((j % NSTG) * KH + wave * PW) * TD_FRAG_BYTES);
#pragma unroll
        for (int i = 0; i < PW; ++i) {
            const char *sp = src + i * TD_FRAG_BYTES;
            asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2" ::"s"(dst + i * TD_FRAG_BYTES), "v"(lane16), "s"(sp) : "memory");
        }
    };
    for (int j = 0; j < LEAD && j < nst; ++j) stage_dma(j);
    int jr = 0;                   // stage the LDS read pointer enters next
    const char *rbase = sm;       // this wave's fragments of the stage being read
    // The read pointer enters stage jr: my pieces of it have landed -- the pieces of the LEAD - 1 stages behind it are
    // younger (vmcnt retires in order) and stay in flight, output stores in between only make the wait stricter --;
    // behind the barrier that holds for every wave, and every wave has consumed stage jr - 2, whose buffer the next
    // pieces go to.
    auto enter = [&]() {
        if (jr + LEAD - 1 < nst) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PW * (LEAD - 1)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (jr + LEAD < nst) stage_dma(jr + LEAD);
        rbase = sm + ((size_t)(jr % NSTG) * KH + kh_half * KHL) * TD_FRAG_BYTES + lane * 16;
        ++jr;
    };
    float *red = (float *)(sm + (size_t)NSTG * KH * TD_FRAG_BYTES) + ((wave / KS) * 64 + lane) * 16;   // KS = 2: the pair's hand-over
    bf16x8 bq[RING];
    enter();
#pragma unroll
    for (int i = 0; i < RING; ++i) bq[i] = *(const bf16x8 *)(rbase + i * TD_FRAG_BYTES);
    __builtin_amdgcn_sched_barrier(0);
    for (int t = t0; t < t1; ++t) {
        const bool more = t + 1 < t1;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // pinned order: left alone, the scheduler sinks every LDS read next to its MFMA (read -> wait -> MFMA)
        td_unrolled<NK>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[i], bq[i % RING], acc, 0, 0, 0);
            constexpr int r = (i + RING) % NK;          // fragment read now: of this tile, or of the next one
            if (i + RING < NK || more) {
                if constexpr (r % KHL == 0) enter();
                bq[i % RING] = *(const bf16x8 *)(rbase + (r % KHL) * TD_FRAG_BYTES);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (KS == 2) {
            // (the next write of `red` lies a whole tile, i.e. NT stage barriers, behind this read)
            if (kh_half == 1) {
#pragma unroll
                for (int r = 0; r < 16; r += 4) *(f32x4 *)(red + r) = f32x4{acc[r], acc[r + 1], acc[r + 2], acc[r + 3]};
            }
            __syncthreads();
            if (kh_half == 0) {
#pragma unroll
                for (int r = 0; r < 16; r += 4) {
                    const f32x4 o = *(const f32x4 *)(red + r);
                    acc[r] += o[0]; acc[r + 1] += o[1]; acc[r + 2] += o[2]; acc[r + 3] += o[3];
                }
            }
        }
        if (rb0 < p.nrb && kh_half == 0) {
            bf16x8 a, b;
#pragma unroll
            for (int j = 0; j < 8; ++j) { a[j] = (__bf16)acc[j]; b[j] = (__bf16)acc[8 + j]; }
            bf16x8 *op = (bf16x8 *)(p.out + ((size_t)t * p.out_frags + p.out_off + 2 * rb) * TD_FRAG_BYTES) + lane;
            op[0] = a;
            op[64] = b;
        }
    }
}

template <int NT, int KH, int KS, int NSTG>
static int td_launch_frag_gemm_res(const TdFragGemmParams &q, hipStream_t stream) {
    const int rbg = td_cdiv(q.nrb, 4 / KS);
    // enough chunks of tiles to give every CU a workgroup, at least four tiles each (the weights are loaded per workgroup)
    int chunks = td_cdiv(TDC_CUS, rbg);
    if (chunks < 1) chunks = 1;
    if (chunks > td_cdiv(q.ntiles, 4)) chunks = td_cdiv(q.ntiles, 4);
    const int per = td_cdiv(q.ntiles, chunks);
    chunks = td_cdiv(q.ntiles, per);
    const size_t shm = (size_t)NSTG * KH * TD_FRAG_BYTES + (KS == 2 ? 2 * 64 * 16 * sizeof(float) : 0);
    TD_HIP(hipFuncSetAttribute((const void *)td_frag_gemm_res_kernel<NT, KH, KS, NSTG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
    hipLaunchKernelGGL((td_frag_gemm_res_kernel<NT, KH, KS, NSTG>), dim3(rbg * 8 * td_cdiv(chunks, 8)), dim3(256), shm, stream, q, per);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

static int td_frag_gemm(const TdFragGemmParams &q, hipStream_t stream) {
    if (q.in_frags == 4 * q.kh && q.kh == 16) {
        if (q.nterm == 4) return td_launch_frag_gemm_res<4, 16, 1, TD_FG_STAGES>(q, stream);
        if (q.nterm == 3) return td_launch_frag_gemm_res<3, 16, 1, TD_FG_STAGES>(q, stream);
    }
    if (q.in_frags == 4 * q.kh && q.kh == 32) {     // hidden in (480, 512]: two waves per row block, four stages of 32 KiB
        if (q.nterm == 4) return td_launch_frag_gemm_res<4, 32, 2, 4>(q, stream);
        if (q.nterm == 3) return td_launch_frag_gemm_res<3, 32, 2, 4>(q, stream);
    }
    hipLaunchKernelGGL(td_frag_gemm_kernel, dim3(td_cdiv(q.nrb, 4), q.ntiles), dim3(256), 0, stream, q);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

// ---------------------------------------------------------------------------------------------
// BPTT sweep of one layer: the whole time loop in ONE launch.  A workgroup owns one batch block of 32
// sequences in one direction; wave u (of 8) owns the 32-unit block u: its d c carry and its
// d h_{t-1} (the accumulator of the recurrent GEMM, which is already in the layout the gate-gradient
// arithmetic works in) stay in its registers across steps.  Per step the waves publish d a_t as MFMA
// B fragments in LDS (double buffered: one barrier per step) and each computes
//     d h_{t-1}[its row blocks] = W_hh^T (A fragments, L2-resident) x d a_t
// d a_t also goes to HBM in both layouts the batched GEMMs after the sweep read (d x = W_ih^T d a, d W).
// ---------------------------------------------------------------------------------------------
// slab gi of d a is the gradient of which bias rows: LSTM i, f, g, o of both; GRU r, z of both, n_x of b_ih, n_h of b_hh
__device__ __forceinline__ void td_bias_add(float *db_ih, float *db_hh, int gru, int H, int gi, int unit, float v) {
    if (!gru) {
        atomicAdd(&db_ih[gi * H + unit], v);
        atomicAdd(&db_hh[gi * H + unit], v);
    } else if (gi < 2) {
        atomicAdd(&db_ih[gi * H + unit], v);
        atomicAdd(&db_hh[gi * H + unit], v);
    } else if (gi == 2) {
        atomicAdd(&db_ih[2 * H + unit], v);
    } else {
        atomicAdd(&db_hh[2 * H + unit], v);
    }
}

struct TdSweepParams {
    TdBpttParams b;
    const char *wt;          // W_hh^T fragments [rb][nterm][kh], direction 0
    size_t wt_dir;           // bytes to direction 1
    int nterm;
    int slab[4];
};

// WIDE (more than 8 unit blocks, rnn_wide.hip's hidden sizes): wave w owns the unit blocks w, w + 8, ... (UPW of
// them), and d a_t (up to 256 KiB per step) is exchanged through the copy that goes to HBM anyway instead of LDS:
// written, fenced, barrier, then read back from L2 as the B operand.
template <int UPW, bool WIDE>
__global__ __launch_bounds__(512) void td_bptt_sweep_kernel(TdSweepParams q) {
    extern __shared__ __attribute__((aligned(16))) char sm[];
    const TdBpttParams &p = q.b;
    // up to 16 unit blocks (hidden <= 512) d a_t (4 KH KiB <= 128 KiB) also fits LDS, single-buffered with a second barrier per
    // step: the recurrent product then reads its B operand there instead of back from L2 (half the L2 traffic of the
    // product, no fence)
    constexpr bool LDSB = WIDE && UPW == 2;
    // WIDE, two directions: a 1-D grid dealt so that XCDs 0-3 work direction 0 and XCDs 4-7 direction 1 (workgroups go to
    // the XCDs round-robin by index): an XCD's 4 MiB L2 then holds ONE direction's W_hh^T (2 MiB at H = 512) instead of both
    int bblk = blockIdx.x, d = blockIdx.y;
    if constexpr (WIDE) {
        if (gridDim.y == 1 && p.dirs == 2) {
            const int xcd = blockIdx.x % 8;
            d = xcd / 4;
            bblk = (blockIdx.x / 8) * 4 + xcd % 4;
            if (bblk >= p.nbb) return;
        }
    }
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63, hf = lane >> 5, m = lane & 31;
    const size_t stage_bytes = (size_t)4 * p.kh * TD_FRAG_BYTES;
    const size_t cols = (size_t)p.L * p.nbb * 32;
    const int hpad = p.kh * 16;
    f32x4 dh[UPW][4], carry[UPW][4];   // accumulator rows 4k .. 4k+3 of each owned unit block
#pragma unroll
    for (int i = 0; i < UPW; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) { dh[i][k] = f32x4{0.f, 0.f, 0.f, 0.f}; carry[i][k] = f32x4{0.f, 0.f, 0.f, 0.f}; }
    // bias gradients = sums of d a over steps and sequences: per lane here, over the 32 sequences at the end
    float dbacc[WIDE ? 1 : 4][16];
    const unsigned slabs = (unsigned)q.slab[0] | (unsigned)q.slab[1] << 4 | (unsigned)q.slab[2] << 8 | (unsigned)q.slab[3] << 12;
    if constexpr (!WIDE) {
#pragma unroll
        for (int gi = 0; gi < 4; ++gi)
#pragma unroll
            for (int r = 0; r < 16; ++r) dbacc[gi][r] = 0.f;
    }

    for (int s = 0; s < p.L; ++s) {
        const int pos = p.L - 1 - s;                   // position in the direction's forward order
        const int l = d ? p.L - 1 - pos : pos;
        const int lprev = d ? l + 1 : l - 1;           // the step processed before l in the forward pass
        const bool has_prev = pos > 0;
        char *dab = WIDE ? p.daf + (size_t)d * p.daf_dir + ((size_t)l * p.nbb + bblk) * stage_bytes
                         : sm + (size_t)(s & 1) * stage_bytes;
#pragma unroll
        for (int i = 0; i < UPW; ++i) {
            const int u = w + 8 * i;
            if (u >= p.nu) continue;
            for (int sidx = 0; sidx < p.n_src; ++sidx) {
                const bf16x8 *gp = (const bf16x8 *)(p.gsrc[sidx] + (((size_t)l * p.nbb + bblk) * p.out_frags + d * p.kh + 2 * u) * TD_FRAG_BYTES) + lane;
                const bf16x8 a = gp[0], b = gp[64];
#pragma unroll
                for (int j = 0; j < 8; ++j) { dh[i][j >> 2][j & 3] += (float)a[j]; dh[i][2 + (j >> 2)][j & 3] += (float)b[j]; }
            }
            const f32x4 *sv = (const f32x4 *)(p.save + (size_t)d * p.save_dir + ((((size_t)l * p.nbb + bblk) * p.nu + u) * 5) * 1024 + lane * 16);
            const f32x4 *svp = (const f32x4 *)(p.save + (size_t)d * p.save_dir + ((((size_t)(has_prev ? lprev : l) * p.nbb + bblk) * p.nu + u) * 5 + 4) * 1024 + lane * 16);
            // two halves of the unit block (accumulator rows 0..7 / 8..15 = one 16-byte fragment row each): d a lives for
            // half a block at a time
#pragma unroll
            for (int half = 0; half < 2; ++half) {
                float da[4][8];
#pragma unroll
                for (int kk = 0; kk < 2; ++kk) {
                    const int k = 2 * half + kk;
                    const f32x4 s0 = sv[k], s1 = sv[256 + k], s2 = sv[512 + k], s3 = sv[768 + k], s4 = sv[1024 + k];
                    const f32x4 pv = has_prev ? svp[k] : f32x4{0.f, 0.f, 0.f, 0.f};   // c_{t-1} (LSTM) / h_{t-1} (GRU)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int r = 4 * kk + e;
                        const float g = dh[i][k][e], prev = pv[e];
                        if (!p.gru) {
                            const float ig = s0[e], fg = s1[e], gg = s2[e], og = s3[e], cv = s4[e];
                            const float tc = td_tanh(cv);
                            const float dcv = carry[i][k][e] + g * og * (1.0f - tc * tc);
                            da[0][r] = dcv * gg * ig * (1.0f - ig);
                            da[1][r] = dcv * prev * fg * (1.0f - fg);
                            da[2][r] = dcv * ig * (1.0f - gg * gg);
                            da[3][r] = g * tc * og * (1.0f - og);
                            carry[i][k][e] = dcv * fg;
                        } else {
                            const float rg = s0[e], zg = s1[e], ng = s2[e], hn = s3[e];
                            const float dht = g + carry[i][k][e];
                            const float dan = dht * (1.0f - zg) * (1.0f - ng * ng);
                            da[0][r] = dan * hn * rg * (1.0f - rg);
                            da[1][r] = dht * (prev - ng) * zg * (1.0f - zg);
                            da[2][r] = dan;
                            da[3][r] = dan * rg;
                            carry[i][k][e] = dht * zg;
                        }
                    }
                }
#pragma unroll
                for (int gi = 0; gi < 4; ++gi) {
                    bf16x8 a;
#pragma unroll
                    for (int j = 0; j < 8; ++j) a[j] = (__bf16)da[gi][j];
                    if constexpr (!WIDE) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) dbacc[gi][8 * half + j] += da[gi][j];
                    }
                    *((bf16x8 *)(p.daf + (size_t)d * p.daf_dir + ((((size_t)l * p.nbb + bblk) * 4 + gi) * p.kh + 2 * u + half) * TD_FRAG_BYTES) + lane) = a;
                    if constexpr (!WIDE) *((bf16x8 *)(dab + ((size_t)gi * p.kh + 2 * u + half) * TD_FRAG_BYTES) + lane) = a;
                    if constexpr (LDSB) *((bf16x8 *)(sm + ((size_t)gi * p.kh + 2 * u + half) * TD_FRAG_BYTES) + lane) = a;
                    __bf16 *pl = p.dapl + (size_t)d * p.dapl_dir + (size_t)gi * hpad * cols + (size_t)l * p.nbb * 32 + bblk * 32 + m;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int unit = 32 * u + (j & 3) + 8 * (2 * half + (j >> 2)) + 4 * hf;
                        pl[(size_t)unit * cols] = a[j];
                    }
                }
            }
        }
        if constexpr (WIDE && !LDSB) __threadfence();   // the fragments must be in L2 before another wave reads them back
        __syncthreads();   // d a_t of every unit block is published (LDS: the other buffer is free again two steps on)
        if (s + 1 < p.L) {
#pragma unroll
            for (int i = 0; i < UPW; ++i) {
                const int u = w + 8 * i;
                if (u >= p.nu) continue;
                const bf16x8 *wp = (const bf16x8 *)(q.wt + (size_t)d * q.wt_dir + (size_t)u * q.nterm * p.kh * TD_FRAG_BYTES) + lane;
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
                if constexpr (WIDE) {
                    // both operands come from L2 (2 KiB per MFMA): the loads run RING fragment pairs ahead of their MFMA,
                    // in a pinned order (left to the scheduler, each pair is loaded right before its use and the sweep
                    // spends its time in L2 latency)
                    // No branch inside the loop (the pair count is a multiple of RING: kh is a multiple of 8 above 8 unit
                    // blocks) and a handful of scalar instructions per pair: with a guarded MFMA and an index / kh per load
                    // the compiler waited vmcnt(0) at every pair and a pair cost 190 issue cycles (td_bptt_step_kernel).
                    constexpr int RING = UPW > 2 ? 4 : 8;
                    const int total = q.nterm * p.kh;
                    const int jump = 2 * p.kh, gap = (q.slab[2] - 2) * p.kh;     // slab map 0, 1, 2 + g, 3 + g (td_layers_backward)
                    const bf16x8 *bsrc = (const bf16x8 *)(LDSB ? sm : dab) + lane;
                    int li = 0;
                    auto load_next = [&](bf16x8 &a, bf16x8 &b) {
                        const int bi = li + (li >= jump ? gap : 0);
                        a = wp[(size_t)li * 64];
                        b = bsrc[(size_t)bi * 64];
                        li = min(li + 1, total - 1);             // the loads past the last pair re-read it
                    };
                    bf16x8 ra[RING], rb[RING];
#pragma unroll
                    for (int j = 0; j < RING; ++j) load_next(ra[j], rb[j]);
                    __builtin_amdgcn_sched_barrier(0);
                    for (int i0 = 0; i0 < total; i0 += RING) {
#pragma unroll
                        for (int j = 0; j < RING; ++j) {
                            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[j], rb[j], acc, 0, 0, 0);
                            load_next(ra[j], rb[j]);
                            __builtin_amdgcn_sched_barrier(0);
                        }
                    }
                } else {
                    for (int term = 0; term < q.nterm; ++term) {
                        const bf16x8 *bp = (const bf16x8 *)(dab + (size_t)q.slab[term] * p.kh * TD_FRAG_BYTES) + lane;
                        const bf16x8 *ap = wp + (size_t)term * p.kh * 64;
#pragma unroll 8
                        for (int ks = 0; ks < p.kh; ++ks) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[ks * 64], bp[ks * 64], acc, 0, 0, 0);
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) dh[i][r >> 2][r & 3] = acc[r];
            }
        }
        if constexpr (LDSB) __syncthreads();   // every wave has read d a_t: the single LDS copy may be overwritten
    }
    if constexpr (!WIDE) {
        if (w < p.nu) {
#pragma unroll
            for (int gi = 0; gi < 4; ++gi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t = td_half_wave_sum(dbacc[gi][r]);
                    const int unit = 32 * w + (r & 3) + 8 * (r >> 2) + 4 * hf;
                    if (m == 0 && unit < p.H) td_bias_add(p.db_ih[d], p.db_hh[d], p.gru, p.H, gi, unit, t);
                }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// The wide sweep with ONE launch per time step (td_use_step_launch, rnn_geom.h; the forward's counterpart is
// td_rnn_step_wide_kernel): a wave owns one (unit block, batch block, direction) of one step, so nu * dirs * batch
// blocks waves spread over the chip instead of 8 waves per (batch block, direction) pulling all of W_hh^T (2 MiB at
// H = 512) through one CU every step (53 us per step at H = 512).  d a of the step before comes back from the copy that
// goes to HBM anyway (a kernel boundary lies in between: no fence), the d c / d h carry of a block lives in `carry`
// ([dir][bblk][u][64 lanes][16] fp32, the forward's state scratch) between launches.  Same order of the MFMAs and the
// same gate-gradient arithmetic as the WIDE sweep above: bit-identical d a.
// ---------------------------------------------------------------------------------------------
// RING: fragment pairs in flight per wave (24 -> 48 KiB), a divisor of the product's nterm * kh pairs, so that the loop has
// no branch: with one (a guarded MFMA, a guarded counter update) the compiler waits vmcnt(0) at every pair and the ring
// degenerates to load -> wait -> MFMA, 190 cycles per pair (the WIDE sweep above still pays that).
template <int RING>
__global__ __launch_bounds__(256, 1) void td_bptt_step_kernel(TdSweepParams q, float *carry_buf, size_t carry_dir, int s) {
    const TdBpttParams &p = q.b;
    const int lane = threadIdx.x & 63, hf = lane >> 5, m = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
    // Workgroups go to the XCDs round-robin by index, and after a kernel boundary an XCD's L2 is cold: what a step reads comes
    // from the Infinity Cache once per XCD that touches it.  An XCD therefore works ONE direction and every G-th unit block
    // of it (G = 8 / dirs XCDs per direction) for all batch blocks: it pulls nu / G slices of W_hh^T and that direction's d a
    // (2.5 MB per step at H = 512, batch 512) instead of slices of both directions and all of d a (4.5 MB).  (Measured at that
    // shape: no difference against the plain (unit block, direction) order -- the step is not bound there; kept for H = 1024.)
    const int xcd = blockIdx.x % 8, slot = blockIdx.x / 8;
    const int G = 8 / p.dirs, ngrp = (p.nbb + nw - 1) / nw;
    const int d = p.dirs == 2 ? xcd / 4 : 0;
    const int u = xcd % G + G * (slot / ngrp);
    const int bblk = (slot % ngrp) * nw + wave;
    if (u >= p.nu || bblk >= p.nbb) return;          // (no barrier in this kernel)
    const size_t stage_bytes = (size_t)4 * p.kh * TD_FRAG_BYTES;
    const size_t cols = (size_t)p.L * p.nbb * 32;
    const int hpad = p.kh * 16;
    const int pos = p.L - 1 - s;                     // position in the direction's forward order
    const int l = d ? p.L - 1 - pos : pos;
    const int lprev = d ? l + 1 : l - 1;             // the step processed before l in the forward pass
    const int lnext = d ? l - 1 : l + 1;             // ... and after it: the sweep's previous step
    const bool has_prev = pos > 0;
    f32x4 dh[4], carry[4];
    float *cp = carry_buf + (size_t)d * carry_dir + (((size_t)bblk * p.nu + u) * 64 + lane) * 16;
    // Everything the gate-gradient arithmetic reads -- the saved activations, c_{t-1}, the carry, the gradient from above --
    // is requested here, ahead of the product: 36 KiB per wave whose latency the product's MFMAs hide (left where they are
    // used, the loads came in 28 dependent groups, each waited for with vmcnt(0): 5 us of a 16 us step).
    const f32x4 *sv = (const f32x4 *)(p.save + (size_t)d * p.save_dir + ((((size_t)l * p.nbb + bblk) * p.nu + u) * 5) * 1024 + lane * 16);
    const f32x4 *svp = (const f32x4 *)(p.save + (size_t)d * p.save_dir + ((((size_t)(has_prev ? lprev : l) * p.nbb + bblk) * p.nu + u) * 5 + 4) * 1024 + lane * 16);
    f32x4 S[5][4], PV[4];
    bf16x8 GA[2], GB[2];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int qn = 0; qn < 5; ++qn) S[qn][k] = sv[256 * qn + k];
        PV[k] = svp[k];
        carry[k] = *(const f32x4 *)(cp + 4 * k);
    }
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
        const bf16x8 *gp = (const bf16x8 *)(p.gsrc[sidx < p.n_src ? sidx : 0] + (((size_t)l * p.nbb + bblk) * p.out_frags + d * p.kh + 2 * u) * TD_FRAG_BYTES) + lane;
        GA[sidx] = gp[0]; GB[sidx] = gp[64];
    }
    __builtin_amdgcn_sched_barrier(0);
    if (s > 0) {
        // d h_t[this unit block] = W_hh^T x d a_{t+1}: both operands from L2, the loads RING fragment pairs ahead of their MFMA
        const bf16x8 *wp = (const bf16x8 *)(q.wt + (size_t)d * q.wt_dir + (size_t)u * q.nterm * p.kh * TD_FRAG_BYTES) + lane;
        const char *dab = p.daf + (size_t)d * p.daf_dir + ((size_t)lnext * p.nbb + bblk) * stage_bytes;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        const int total = q.nterm * p.kh;            // a multiple of RING (td_launch_bptt_steps)
        // pair i: W_hh^T fragment i of the unit block against d a fragment i, or i + gap from the third term on (the GRU's
        // slabs r, z, n_h: td_launch_bptt_steps checks that the slab map has this form).  A handful of scalar instructions
        // per pair: with (term, k-step) bookkeeping the pair cost 30 instructions = 150 issue cycles against the MFMA's 32.
        const int jump = 2 * p.kh, gap = (q.slab[2] - 2) * p.kh;
        int li = 0;
        const bf16x8 *dabl = (const bf16x8 *)dab + lane;
        auto load_next = [&](bf16x8 &a, bf16x8 &b) {
            const int bi = li + (li >= jump ? gap : 0);
            a = wp[(size_t)li * 64];
            b = dabl[(size_t)bi * 64];
            li = min(li + 1, total - 1);             // the loads past the last pair re-read it
        };
        bf16x8 ra[RING], rb[RING];
#pragma unroll
        for (int j = 0; j < RING; ++j) load_next(ra[j], rb[j]);
        __builtin_amdgcn_sched_barrier(0);
        for (int i0 = 0; i0 < total; i0 += RING) {
#pragma unroll
            for (int j = 0; j < RING; ++j) {
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ra[j], rb[j], acc, 0, 0, 0);
                load_next(ra[j], rb[j]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) dh[r >> 2][r & 3] = acc[r];
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) { dh[k] = f32x4{0.f, 0.f, 0.f, 0.f}; carry[k] = f32x4{0.f, 0.f, 0.f, 0.f}; }   // (the scratch holds the last sweep's)
    }
#pragma unroll
    for (int sidx = 0; sidx < 2; ++sidx) {
        if (sidx < p.n_src) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { dh[j >> 2][j & 3] += (float)GA[sidx][j]; dh[2 + (j >> 2)][j & 3] += (float)GB[sidx][j]; }
        }
    }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        float da[4][8];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            const int k = 2 * half + kk;
            const f32x4 s0 = S[0][k], s1 = S[1][k], s2 = S[2][k], s3 = S[3][k], s4 = S[4][k];
            const f32x4 pv = has_prev ? PV[k] : f32x4{0.f, 0.f, 0.f, 0.f};   // c_{t-1} (LSTM) / h_{t-1} (GRU)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = 4 * kk + e;
                const float g = dh[k][e], prev = pv[e];
                if (!p.gru) {
                    const float ig = s0[e], fg = s1[e], gg = s2[e], og = s3[e], cv = s4[e];
                    const float tc = td_tanh(cv);
                    const float dcv = carry[k][e] + g * og * (1.0f - tc * tc);
                    da[0][r] = dcv * gg * ig * (1.0f - ig);
                    da[1][r] = dcv * prev * fg * (1.0f - fg);
                    da[2][r] = dcv * ig * (1.0f - gg * gg);
                    da[3][r] = g * tc * og * (1.0f - og);
                    carry[k][e] = dcv * fg;
                } else {
                    const float rg = s0[e], zg = s1[e], ng = s2[e], hn = s3[e];
                    const float dht = g + carry[k][e];
                    const float dan = dht * (1.0f - zg) * (1.0f - ng * ng);
                    da[0][r] = dan * hn * rg * (1.0f - rg);
                    da[1][r] = dht * (prev - ng) * zg * (1.0f - zg);
                    da[2][r] = dan;
                    da[3][r] = dan * rg;
                    carry[k][e] = dht * zg;
                }
            }
        }
#pragma unroll
        for (int gi = 0; gi < 4; ++gi) {
            bf16x8 a;
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = (__bf16)da[gi][j];
            *((bf16x8 *)(p.daf + (size_t)d * p.daf_dir + ((((size_t)l * p.nbb + bblk) * 4 + gi) * p.kh + 2 * u + half) * TD_FRAG_BYTES) + lane) = a;
            __bf16 *pl = p.dapl + (size_t)d * p.dapl_dir + (size_t)gi * hpad * cols + (size_t)l * p.nbb * 32 + bblk * 32 + m;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int unit = 32 * u + (j & 3) + 8 * (2 * half + (j >> 2)) + 4 * hf;
                pl[(size_t)unit * cols] = a[j];
            }
        }
    }
    if (s + 1 < p.L) {
#pragma unroll
        for (int k = 0; k < 4; ++k) *(f32x4 *)(cp + 4 * k) = carry[k];
    }
}

static int td_launch_bptt_steps(const TdSweepParams &q, float *carry, size_t carry_dir, hipStream_t stream) {
    const TdBpttParams &p = q.b;
    int nw = p.nu * p.dirs * p.nbb / 256;          // waves per workgroup: spread the tasks over the CUs first
    nw = nw < 1 ? 1 : (nw > 4 ? 4 : nw);
    const int grid = 8 * td_cdiv(p.nu, 8 / p.dirs) * td_cdiv(p.nbb, nw);
    const int total = q.nterm * p.kh;
    for (int term = 0; term < q.nterm; ++term)
        TD_REQUIRE(q.slab[term] == term + (term >= 2 ? q.slab[2] - 2 : 0) && q.slab[2] >= 2, "per-step sweep: unexpected gate-slab map");
    for (int s = 0; s < p.L; ++s) {
        if (total % 24 == 0) hipLaunchKernelGGL(td_bptt_step_kernel<24>, dim3(grid), dim3(64 * nw), 0, stream, q, carry, carry_dir, s);
        else if (total % 16 == 0) hipLaunchKernelGGL(td_bptt_step_kernel<16>, dim3(grid), dim3(64 * nw), 0, stream, q, carry, carry_dir, s);
        else hipLaunchKernelGGL(td_bptt_step_kernel<8>, dim3(grid), dim3(64 * nw), 0, stream, q, carry, carry_dir, s);
    }
    TD_LAUNCH_CHECK();
    return TD_OK;
}

// ---------------------------------------------------------------------------------------------
// weight-gradient GEMM: C[m][col(n)] += sum_k A[m][k] B[n][k]   (bf16 operands with k contiguous, fp32 accumulation)
//
// d W = d a . [x ; h]^T over all L * batch columns: M x N of a few hundred to a few thousand, K in the tens to hundreds
// of thousands.  A workgroup (4 waves, 2 x 2) owns a 256 x 256 tile of C and one chunk of K (split K: with 4 x 2 tiles
// only the chunks fill the chip); a stage = 64 columns of both operands, brought in by 16-byte loads of whole 128-byte row
// segments, parked in registers while the previous stage is multiplied, and written to one of two LDS buffers whose rows
// are padded to 144 bytes (the 32 rows a fragment read touches then fall into 16 distinct bank quads).  Both operands
// have k contiguous, which is exactly the lane layout of v_mfma_f32_32x32x16_bf16's A and B fragments (8 k values per
// lane), so a fragment is one ds_read_b128.  A chunk's tile goes to its own slab of `partial` and td_dw_reduce_kernel
// adds the slabs in order: the sums do not depend on which workgroup finishes first (no atomics).
// ---------------------------------------------------------------------------------------------
struct TdDwParams {
    const __bf16 *A, *B;     // row-major, k contiguous
    float *C;
    size_t lda, ldb;
    int ldc, M, N, K;        // M rows of A, N rows of B (a multiple of 32)
    int hp, hv;              // column map: n -> (n / hp) * hv + n % hp, valid when n % hp < hv
    // row map: row m of the product is row m % mp of the slab a_slab[m / mp] (elements from A) and goes to row
    // (m / mp) * mv + m % mp of C, valid when m % mp < mv: the gates' slabs of d a (padded to whole unit blocks, and for the
    // GRU's W_hh not adjacent) in ONE product.  mp = mv = M, a_slab[0] = 0: no map.
    int mp, mv;
    size_t a_slab[4];
    int kc, nsplit;          // columns per chunk (a multiple of TD_DW_BK), chunks
    float *partial;          // nsplit > 1: [chunk][M][N] fp32
};

#define TD_DW_BK 64
#define TD_DW_ROWB 144

// WM x WN: 32 x 32 MFMA tiles per wave; the four waves sit 2 x 2, so the workgroup's tile is (64 WM) x (64 WN)
// EDGE: M, N or the chunk length are not multiples of the tile: loads and stores are predicated
template <int WM, int WN, bool EDGE>
__global__ __launch_bounds__(256, 1) void td_dw_gemm_kernel(TdDwParams p) {
    constexpr int BM = 64 * WM, BN = 64 * WN, BK = TD_DW_BK, ROWB = TD_DW_ROWB;
    constexpr int NLA = BM / 32, NLB = BN / 32;      // 16-byte pieces per thread and stage: rows row0 + 32 i
    __shared__ __attribute__((aligned(16))) char lds[2 * (BM + BN) * ROWB];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, r = lane & 31, hf = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN, split = blockIdx.z;
    const int k0 = split * p.kc, k1 = min(p.K, k0 + p.kc);
    const int c = tid & 7, row0 = tid >> 3;
    const __bf16 *ag = p.A + (size_t)(m0 + row0) * p.lda + c * 8, *bg = p.B + (size_t)(n0 + row0) * p.ldb + c * 8;
    const size_t astep = (size_t)32 * p.lda, bstep = (size_t)32 * p.ldb;
    // EDGE: this thread's rows of A through the row map (a row beyond M or in a slab's padding reads as zero)
    const __bf16 *arow[EDGE ? NLA : 1];
    if constexpr (EDGE) {
#pragma unroll
        for (int i = 0; i < NLA; ++i) {
            const int m = m0 + row0 + 32 * i;
            arow[i] = (m < p.M && m % p.mp < p.mv) ? p.A + p.a_slab[m / p.mp] + (size_t)(m % p.mp) * p.lda + c * 8 : nullptr;
        }
    }
    bf16x8 va[NLA], vb[NLB];
    auto gload = [&](int k) {
        if constexpr (EDGE) {
            const bool inside = k + c * 8 < k1;      // K is a multiple of 8: a piece is wholly inside or outside
            const bf16x8 zero = {};
#pragma unroll
            for (int i = 0; i < NLA; ++i) va[i] = (inside && arow[i]) ? *(const bf16x8 *)(arow[i] + k) : zero;
#pragma unroll
            for (int i = 0; i < NLB; ++i) vb[i] = (inside && n0 + row0 + 32 * i < p.N) ? *(const bf16x8 *)(bg + i * bstep + k) : zero;
        } else {
#pragma unroll
            for (int i = 0; i < NLA; ++i) va[i] = *(const bf16x8 *)(ag + i * astep + k);
#pragma unroll
            for (int i = 0; i < NLB; ++i) vb[i] = *(const bf16x8 *)(bg + i * bstep + k);
        }
    };
    auto lstore = [&](int buf) {
        char *base = lds + buf * (BM + BN) * ROWB + c * 16;
#pragma unroll
        for (int i = 0; i < NLA; ++i) *(bf16x8 *)(base + (row0 + 32 * i) * ROWB) = va[i];
#pragma unroll
        for (int i = 0; i < NLB; ++i) *(bf16x8 *)(base + (BM + row0 + 32 * i) * ROWB) = vb[i];
    };
    f32x16 acc[WM][WN];
#pragma unroll
    for (int i = 0; i < WM; ++i)
#pragma unroll
        for (int j = 0; j < WN; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    if (k0 < k1) {
        gload(k0);
        lstore(0);
    }
    __syncthreads();
    int buf = 0;
    for (int k = k0; k < k1; k += BK) {
        const bool more = k + BK < k1;
        if (more) gload(k + BK);
        const char *ab = lds + buf * (BM + BN) * ROWB + (wm * 32 * WM + r) * ROWB + hf * 16;
        const char *bb = lds + buf * (BM + BN) * ROWB + (BM + wn * 32 * WN + r) * ROWB + hf * 16;
        // the fragments of k-step ks + 1 are read while k-step ks is multiplied (left to itself the compiler reads a fragment,
        // waits for it and issues the four matrix instructions that use it: 65 % of this kernel's rate)
        bf16x8 fa[2][WM], fb[2][WN];
#pragma unroll
        for (int i = 0; i < WM; ++i) fa[0][i] = *(const bf16x8 *)(ab + i * 32 * ROWB);
#pragma unroll
        for (int j = 0; j < WN; ++j) fb[0][j] = *(const bf16x8 *)(bb + j * 32 * ROWB);
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            if (ks + 1 < BK / 16) {
#pragma unroll
                for (int i = 0; i < WM; ++i) fa[(ks + 1) & 1][i] = *(const bf16x8 *)(ab + i * 32 * ROWB + (ks + 1) * 32);
#pragma unroll
                for (int j = 0; j < WN; ++j) fb[(ks + 1) & 1][j] = *(const bf16x8 *)(bb + j * 32 * ROWB + (ks + 1) * 32);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < WM; ++i)
#pragma unroll
                for (int j = 0; j < WN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][i], fb[ks & 1][j], acc[i][j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (more) lstore(buf ^ 1);
        __syncthreads();     // the other buffer is complete, and every wave is done with this one
        buf ^= 1;
    }
    // accumulator element q of lane (r, hf): row (q & 3) + 8 (q >> 2) + 4 hf, column r of its 32 x 32 tile.  A slab of the row
    // map is a whole number of 32-row tiles (or the map is the identity), so a tile's rows share one slab: one division a tile.
#pragma unroll
    for (int i = 0; i < WM; ++i) {
        const int mb = m0 + (wm * WM + i) * 32;
        const int rb = mb % p.mp + 4 * hf;              // row within the slab of accumulator element 0
        const size_t crow0 = (size_t)(mb / p.mp) * p.mv + rb;
        const int mlim = min(p.M - mb - 4 * hf, p.mv - rb);   // element q is valid while (q & 3) + 8 (q >> 2) is below this
#pragma unroll
        for (int j = 0; j < WN; ++j) {
            const int n = n0 + (wn * WN + j) * 32 + r;
            if (n >= p.N) continue;
            if (p.nsplit > 1) {
                float *dst = p.partial + (size_t)split * p.M * p.N + (size_t)(mb + 4 * hf) * p.N + n;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int dm = (q & 3) + 8 * (q >> 2);
                    if (dm < p.M - mb - 4 * hf) dst[(size_t)dm * p.N] = acc[i][j][q];
                }
            } else if (n % p.hp < p.hv) {
                float *dst = p.C + crow0 * p.ldc + (n / p.hp) * p.hv + n % p.hp;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const int dm = (q & 3) + 8 * (q >> 2);
                    if (dm < mlim) dst[(size_t)dm * p.ldc] += acc[i][j][q];     // this workgroup owns the tile
                }
            }
        }
    }
}

// C[m][col(n)] += sum over the chunks' slabs (each M x N, packed), in chunk order
__global__ __launch_bounds__(256) void td_dw_reduce_kernel(TdDwParams p) {
    const size_t total = (size_t)p.M * p.N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int n = (int)(i % p.N), m = (int)(i / p.N);
        if (n % p.hp >= p.hv || m % p.mp >= p.mv) continue;
        float s = 0.f;
        for (int k = 0; k < p.nsplit; ++k) s += p.partial[(size_t)k * total + i];
        p.C[(size_t)((m / p.mp) * p.mv + m % p.mp) * p.ldc + (n / p.hp) * p.hv + n % p.hp] += s;
    }
}

// Chunks of K: enough of them that tiles x chunks cover the chip, each at least TD_DW_MIN_KC columns, all slabs within the
// workspace's TD_DW_PARTIAL_BYTES.  Tile: 256 x 256 (a wave 128 x 128: 8 fragment reads feed 16 matrix instructions);
// 256 x 64 where B has few rows (the first layer's input channels).
// Small products (the narrow widths: a few hundred rows and columns): 128 x 128 tiles, so that tiles x chunks still cover
// the chip and a workgroup does not multiply a mostly empty 256 x 256 tile (37 us per call at H = 92 that way).
#define TD_DW_MIN_KC 512
static int td_dw_gemm(TdDwParams q, float *partial, hipStream_t stream) {
    // unpadded, adjacent slabs are plain rows: no map (and with it the unpredicated kernels where the shape allows them)
    bool plain_rows = q.mv == q.mp;
    for (int gi = 0; gi < q.M / q.mp && plain_rows; ++gi) plain_rows = q.a_slab[gi] == (size_t)gi * q.mp * q.lda;
    if (plain_rows) {
        q.mp = q.mv = q.M;
        q.a_slab[0] = 0;
    }
    const bool narrow = q.N <= 64;
    const bool small = !narrow && q.M <= 512 && q.N <= 512;
    const int bm = small ? 128 : 256, bn = narrow ? 64 : (small ? 128 : 256);
    const int tm = td_cdiv(q.M, bm), tn = td_cdiv(q.N, bn);
    int ns = 1;
    while (partial && ns < 128 && tm * tn * ns < 256 && q.K / (2 * ns) >= TD_DW_MIN_KC &&
           (size_t)q.M * q.N * sizeof(float) * (2 * ns) <= TD_DW_PARTIAL_BYTES)
        ns *= 2;
    q.kc = TD_DW_BK * td_cdiv(td_cdiv(q.K, ns), TD_DW_BK);
    q.nsplit = td_cdiv(q.K, q.kc);
    q.partial = partial;
    TD_REQUIRE(q.K % 8 == 0 && q.lda % 8 == 0 && q.ldb % 8 == 0, "weight-gradient GEMM: K and the leading dimensions must be multiples of 8");
    const bool edge = q.M % bm != 0 || q.N % bn != 0 || q.K % TD_DW_BK != 0 || q.mp != q.M;   // a short last chunk is whole stages still
    const dim3 grid(tm, tn, q.nsplit);
    if (small) hipLaunchKernelGGL((td_dw_gemm_kernel<2, 2, true>), grid, dim3(256), 0, stream, q);
    else if (narrow && edge) hipLaunchKernelGGL((td_dw_gemm_kernel<4, 1, true>), grid, dim3(256), 0, stream, q);
    else if (narrow) hipLaunchKernelGGL((td_dw_gemm_kernel<4, 1, false>), grid, dim3(256), 0, stream, q);
    else if (edge) hipLaunchKernelGGL((td_dw_gemm_kernel<4, 4, true>), grid, dim3(256), 0, stream, q);
    else hipLaunchKernelGGL((td_dw_gemm_kernel<4, 4, false>), grid, dim3(256), 0, stream, q);
    TD_LAUNCH_CHECK();
    if (q.nsplit > 1) {
        const size_t total = (size_t)q.M * q.N;
        hipLaunchKernelGGL(td_dw_reduce_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, q);
        TD_LAUNCH_CHECK();
    }
    return TD_OK;
}

// d b[m] += sum_k A[m][k] (one wave per row), added to up to two destinations
__global__ __launch_bounds__(256) void td_bias_reduce_kernel(const __bf16 *A, size_t lda, int M, size_t K, float *dst0, float *dst1) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int m = blockIdx.x * 4 + wave;
    if (m >= M) return;
    const __bf16 *row = A + (size_t)m * lda;
    float s = 0.f;
    for (size_t k = (size_t)lane * 8; k < K; k += 512) {
        const bf16x8 v = *(const bf16x8 *)(row + k);
#pragma unroll
        for (int j = 0; j < 8; ++j) s += (float)v[j];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    if (lane == 0) {
        if (dst0) atomicAdd(&dst0[m], s);
        if (dst1) atomicAdd(&dst1[m], s);
    }
}

// ---------------------------------------------------------------------------------------------
// backward entry point
// A second stream of the library's own for the weight-gradient GEMMs.  At the reference's batch the BPTT sweep of a layer keeps
// 32 of the 256 CUs busy for 0.8 ms; the GEMMs of the layer above (which read only that layer's d a and the activations the
// forward pass saved) need nothing from it and nothing needs them before the optimizer step, so they run beside it: forked
// behind an event after the layer's sweep, joined at the end of the backward pass.  A second side stream carries the head's
// parameter gradients and the time-embedding MLP's backward (td_backward_impl): at the narrow widths the side work is longer
// than the sweeps (MIMIC-III shape, batch 256: 270 us of GEMMs + 280 us of head / time-MLP kernels against 260 us of sweeps),
// so the two kinds run beside each other as well.  Two streams and 19 events per device, created at the first use (the same
// forks / joins are captured as such when the caller's stream is under graph capture).
#ifndef TD_SIDE_GEMM_MAX_CUS
#define TD_SIDE_GEMM_MAX_CUS 64     // CUs a sweep occupies (measured at H = 256: 32 CUs +3.7 %, 128 CUs -7 %)
#endif
struct TdSideStream {
    hipStream_t stream, stream2;            // weight-gradient GEMMs; head parameter gradients + time-embedding backward
    hipEvent_t fork[17], join, join2;       // one fork per layer + the head
    bool ready;
};
static TdSideStream *td_side_stream(void) {
    static TdSideStream side[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
    TdSideStream &s = side[dev];
    if (!s.ready) {
        if (hipStreamCreateWithFlags(&s.stream, hipStreamNonBlocking) != hipSuccess) return nullptr;
        if (hipStreamCreateWithFlags(&s.stream2, hipStreamNonBlocking) != hipSuccess) return nullptr;
        for (int i = 0; i < 17; ++i)
            if (hipEventCreateWithFlags(&s.fork[i], hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&s.join, hipEventDisableTiming) != hipSuccess) return nullptr;
        if (hipEventCreateWithFlags(&s.join2, hipEventDisableTiming) != hipSuccess) return nullptr;
        s.ready = true;
    }
    return &s;
}

// ---------------------------------------------------------------------------------------------
// The recurrent layers' part of the backward pass, top down: the gradient w.r.t. the top layer's output is in gout[N-1][0]
// (fragments) when this is called -- written by the denoiser's head backward, or by the classifier's (td_cls_backward).
static TdSideStream *td_side_for(const TdRnnGeom &g, int nbb, td_layer_done_fn layer_done) {
    // where a sweep leaves most of the chip idle, and not with a per-layer callback (the data-parallel trainer starts a
    // layer's all-reduce from it, behind an event on the caller's stream)
    if (layer_done || td_side_gemm_mode() == 0 || nbb * g.dirs * g.nu > TD_SIDE_GEMM_MAX_CUS) return nullptr;
    return td_side_stream();
}

// `forked`: the caller has already put work on the second side stream (the head's parameter gradients): it is joined here too
static int td_layers_backward(const td_rnn_config *cfg, const TdRnnGeom &g, const void *packT, int batch, int L, char *ws,
                              float *const *grad_layer, td_layer_done_fn layer_done, void *user, hipStream_t stream,
                              TdSideStream *forked = nullptr) {
    const int N = cfg->layers, D = g.dirs, H = cfg->hidden, G = cfg->gru ? 3 : 4, DH = D * H;
    const TdTrainLayout t = td_train_layout(cfg, g, batch, L);
    const TdPackTGeom tg = td_packT_geom(cfg, g);
    int rc = TD_OK;
    float *dwpart = t.cols >= 4 * 512 ? (float *)(ws + t.dwpart) : nullptr;
    // The weight-gradient GEMMs beside the sweeps below them (td_side_stream)
    TdSideStream *side = forked ? forked : td_side_for(g, t.nbb, layer_done);
    hipStream_t gstream = side ? side->stream : stream;
    // ---- layers, top down ------------------------------------------------------------------------
    const int KT = D * g.kh;
    for (int l = N - 1; l >= 0; --l) {
        const int n_src = l == N - 1 ? 1 : D;
        const size_t packT_dir = tg.hh_bytes + (l > 0 ? tg.ih_bytes : 0);
        const char *wT = (const char *)packT + tg.layer_off[l];
        bool bias_in_sweep = g.nu <= 8;       // the sweeps of up to 8 unit blocks fold the bias gradients in (not the per-step form)
        {
            TdSweepParams q;
            TdBpttParams &p = q.b;
            p.gsrc[0] = ws + t.gout[l][0]; p.gsrc[1] = ws + t.gout[l][1]; p.n_src = n_src;
            p.save = (const float *)(ws + t.save[l]); p.save_dir = t.save_dir_floats;
            p.daf = ws + t.daf[l][0]; p.daf_dir = D > 1 ? t.daf[l][1] - t.daf[l][0] : 0;
            p.dapl = (__bf16 *)(ws + t.dapl[l][0]); p.dapl_dir = D > 1 ? (t.dapl[l][1] - t.dapl[l][0]) / 2 : 0;
            p.L = L; p.nbb = t.nbb; p.nu = g.nu; p.kh = g.kh; p.dirs = D; p.out_frags = KT; p.gru = cfg->gru; p.H = H;
            for (int d = 0; d < 2; ++d) {
                float *const *gw = grad_layer + (size_t)(l * D + (d < D ? d : 0)) * 4;
                TD_REQUIRE(gw[2] && gw[3], "NULL bias gradient pointer for layer %d dir %d", l, d);
                p.db_ih[d] = gw[2]; p.db_hh[d] = gw[3];
            }
            q.wt = wT; q.wt_dir = packT_dir; q.nterm = G;
            q.slab[0] = 0; q.slab[1] = 1; q.slab[2] = cfg->gru ? 3 : 2; q.slab[3] = 3;
            const int mode = td_layer_kernel_mode();
            const int prof_slot = l > 0 ? td_profile_begin(TD_PROFILE_BPTT_SWEEP, stream) : -1;
            if ((mode == TD_LAYER_KERNEL_AUTO || mode == TD_LAYER_KERNEL_CLUSTER) && td_bptt_cluster_supported(g, t.nbb, L)) {
                TdClusterBwdParams cp;
                cp.gsrc[0] = p.gsrc[0]; cp.gsrc[1] = p.gsrc[1]; cp.n_src = n_src;
                cp.save = p.save; cp.save_dir = p.save_dir; cp.daf = p.daf; cp.daf_dir = p.daf_dir;
                cp.dapl = p.dapl; cp.dapl_dir = p.dapl_dir; cp.wt = wT; cp.wt_dir = packT_dir;
                cp.inbox = (unsigned long long *)(ws + t.inbox);
                cp.L = L; cp.nbb = t.nbb; cp.dirs = D; cp.out_frags = KT; cp.H = H;
                for (int d = 0; d < 2; ++d) { cp.db_ih[d] = p.db_ih[d]; cp.db_hh[d] = p.db_hh[d]; }
                rc = td_launch_bptt_cluster(cfg->gru, g, cp, stream);
                if (rc != TD_OK) return rc;
            } else if (mode != TD_LAYER_KERNEL_BATCH && td_use_step_launch(g.nu, t.nbb, D)) {
                bias_in_sweep = false;
                rc = td_launch_bptt_steps(q, (float *)(ws + t.dc[0]), (t.dc[1] - t.dc[0]) / sizeof(float), stream);
                if (rc != TD_OK) return rc;
            } else if (g.nu <= 8) {
                const size_t shm = (size_t)2 * 4 * g.kh * TD_FRAG_BYTES;
                TD_HIP(hipFuncSetAttribute((const void *)td_bptt_sweep_kernel<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
                hipLaunchKernelGGL((td_bptt_sweep_kernel<1, false>), dim3(t.nbb, D), dim3(512), shm, stream, q);
            } else if (g.nu <= 16) {
                const size_t shm = (size_t)4 * g.kh * TD_FRAG_BYTES;
                TD_HIP(hipFuncSetAttribute((const void *)td_bptt_sweep_kernel<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
                hipLaunchKernelGGL((td_bptt_sweep_kernel<2, true>), D == 2 ? dim3(8 * td_cdiv(t.nbb, 4)) : dim3(t.nbb, 1), dim3(512), shm, stream, q);
            } else {
                hipLaunchKernelGGL((td_bptt_sweep_kernel<4, true>), D == 2 ? dim3(8 * td_cdiv(t.nbb, 4)) : dim3(t.nbb, 1), dim3(512), 0, stream, q);
            }
            td_profile_end(prof_slot, stream);
            TD_LAUNCH_CHECK();
        }
        if (side) {     // this layer's d a is complete: its GEMMs may start
            TD_HIP(hipEventRecord(side->fork[l], stream));
            TD_HIP(hipStreamWaitEvent(side->stream, side->fork[l], 0));
        }
        // gradient for the layer below: d x = W_ih^T d a, all steps at once
        if (l > 0) {
            for (int d = 0; d < D; ++d) {
                TdFragGemmParams q;
                q.wt = wT + (size_t)d * packT_dir + tg.hh_bytes;
                q.in = ws + t.daf[l][d]; q.in_tile_off = 0; q.in_frags = 4 * g.kh;
                q.out = ws + t.gout[l - 1][d]; q.out_frags = KT; q.out_off = 0;
                q.ntiles = L * t.nbb; q.nrb = D * g.nu; q.nterm = G; q.kh = g.kh;
                q.slab[0] = 0; q.slab[1] = 1; q.slab[2] = 2; q.slab[3] = 3;
                if ((rc = td_frag_gemm(q, stream)) != TD_OK) return rc;
                // d x is the gradient w.r.t. the DROPPED output of the layer below: the same mask takes it to that output
                if (td_dropout_p() > 0.f) {
                    rc = td_dropout_apply(ws + t.gout[l - 1][d], t.act_bytes, l - 1, stream);
                    if (rc != TD_OK) return rc;
                }
            }
        }
        // weight and bias gradients
        for (int d = 0; d < D; ++d) {
            float *const *gw = grad_layer + (size_t)(l * D + d) * 4;
            TD_REQUIRE(gw[0] && gw[1] && gw[2] && gw[3], "NULL gradient pointer for layer %d dir %d", l, d);
            const __bf16 *dapl = (const __bf16 *)(ws + t.dapl[l][d]);
            const int In = l == 0 ? cfg->c_in : DH;
            // LSTM with an unpadded width: the four gates' slabs of d a and their rows of d W are contiguous, so all
            // gates go in one GEMM / one bias reduction per tensor (GM = 4 gates per call)
            const int GM = (!cfg->gru && g.hpad == H) ? G : 1;
            {
                // every gate in ONE product per tensor: the gates' slabs of d a (hpad rows each, H of them valid; for the GRU's
                // W_hh the slabs r, z, r * n) are the row blocks of A, mapped to the gates' rows of d W
                TdDwParams q;
                q.M = G * g.hpad; q.mp = g.hpad; q.mv = H;
                for (int gate = 0; gate < G; ++gate) q.a_slab[gate] = (size_t)gate * t.dapl_slab_elems;
                q.a_slab[3] = G == 4 ? q.a_slab[3] : 0;
                // W_ih
                q.A = dapl; q.lda = t.cols;
                q.B = (const __bf16 *)(ws + t.plainx[l]); q.ldb = t.cols;     // this layer's input (the dropped copy, with dropout)
                q.C = gw[0]; q.ldc = In; q.K = (int)t.cols;
                if (l == 0) { q.N = t.cpad; q.hp = t.cpad; q.hv = cfg->c_in; }
                else { q.N = D * g.hpad; q.hp = g.hpad; q.hv = H; }
                rc = td_dw_gemm(q, dwpart, gstream);
                if (rc != TD_OK) return rc;
                // W_hh: h_{t-1} is this direction's own output one step earlier in its processing order
                if (L > 1) {
                    const __bf16 *hpl = (const __bf16 *)(ws + t.plain[l + 1]) + (size_t)d * g.hpad * t.cols;
                    if (cfg->gru) q.a_slab[2] = (size_t)3 * t.dapl_slab_elems;
                    q.A = d == 0 ? dapl + t.Bp : dapl;
                    q.B = d == 0 ? hpl : hpl + t.Bp;
                    q.C = gw[1]; q.ldc = H; q.K = (int)(t.cols - t.Bp);
                    q.N = g.hpad; q.hp = g.hpad; q.hv = H;
                    rc = td_dw_gemm(q, dwpart, gstream);
                    if (rc != TD_OK) return rc;
                }
            }
            for (int gate = 0; gate < G; gate += GM) {
                const int slab_x = gate, slab_h = (cfg->gru && gate == 2) ? 3 : gate;
                // biases (hidden sizes up to 256: folded into the sweep)
                if (bias_in_sweep) continue;
                hipLaunchKernelGGL(td_bias_reduce_kernel, dim3(td_cdiv(GM * H, 4)), dim3(256), 0, stream,
                                   dapl + (size_t)slab_x * t.dapl_slab_elems, t.cols, GM * H, t.cols, gw[2] + gate * H,
                                   slab_h == slab_x ? gw[3] + gate * H : (float *)nullptr);
                TD_LAUNCH_CHECK();
                if (slab_h != slab_x) {
                    hipLaunchKernelGGL(td_bias_reduce_kernel, dim3(td_cdiv(H, 4)), dim3(256), 0, stream,
                                       dapl + (size_t)slab_h * t.dapl_slab_elems, t.cols, H, t.cols, gw[3] + gate * H, (float *)nullptr);
                    TD_LAUNCH_CHECK();
                }
            }
        }
        if (layer_done) layer_done(l, user);   // every kernel that writes layer l's parameter gradients is enqueued
    }
    if (side) {
        TD_HIP(hipEventRecord(side->join, side->stream));
        TD_HIP(hipStreamWaitEvent(stream, side->join, 0));
        if (forked) {
            TD_HIP(hipEventRecord(side->join2, side->stream2));
            TD_HIP(hipStreamWaitEvent(stream, side->join2, 0));
        }
    }
    return TD_OK;
}

extern "C" int td_rnn_backward(const td_rnn_config *cfg, const void *packT, const float *const *misc, const float *film,
                               const float *grad_y, int32_t batch, int32_t seq_len, void *workspace,
                               float *const *grad_layer, float *const *grad_misc, float *grad_film, void *stream_) {
    return td_rnn_backward_cb(cfg, packT, misc, film, grad_y, batch, seq_len, workspace, grad_layer, grad_misc, grad_film,
                              nullptr, nullptr, stream_);
}

extern "C" int td_rnn_time_film_backward(const td_rnn_config *cfg, const float *const *misc, const int64_t *t, int32_t n_t,
                                         const float *grad_film, float *const *grad_misc, void *workspace, void *stream_);   // rnn.hip

static int td_backward_impl(const td_rnn_config *cfg, const void *packT, const float *const *misc, const float *film,
                            const float *grad_y, int32_t batch, int32_t seq_len, void *workspace, float *const *grad_layer,
                            float *const *grad_misc, float *grad_film, const int64_t *t_steps, void *film_ws,
                            td_layer_done_fn layer_done, void *user, void *stream_);

extern "C" int td_rnn_backward_cb(const td_rnn_config *cfg, const void *packT, const float *const *misc, const float *film,
                                  const float *grad_y, int32_t batch, int32_t seq_len, void *workspace,
                                  float *const *grad_layer, float *const *grad_misc, float *grad_film,
                                  td_layer_done_fn layer_done, void *user, void *stream_) {
    return td_backward_impl(cfg, packT, misc, film, grad_y, batch, seq_len, workspace, grad_layer, grad_misc, grad_film, nullptr,
                            nullptr, layer_done, user, stream_);
}

extern "C" int td_rnn_backward_all(const td_rnn_config *cfg, const void *packT, const float *const *misc, const float *film,
                                   const int64_t *t, const float *grad_y, int32_t batch, int32_t seq_len, void *workspace,
                                   void *film_workspace, float *const *grad_layer, float *const *grad_misc, float *grad_film,
                                   td_layer_done_fn layer_done, void *user, void *stream_) {
    TD_REQUIRE(t && film_workspace, "NULL argument");
    return td_backward_impl(cfg, packT, misc, film, grad_y, batch, seq_len, workspace, grad_layer, grad_misc, grad_film, t,
                            film_workspace, layer_done, user, stream_);
}

static int td_backward_impl(const td_rnn_config *cfg, const void *packT, const float *const *misc, const float *film,
                            const float *grad_y, int32_t batch, int32_t seq_len, void *workspace, float *const *grad_layer,
                            float *const *grad_misc, float *grad_film, const int64_t *t_steps, void *film_ws,
                            td_layer_done_fn layer_done, void *user, void *stream_) {
    TD_REQUIRE(cfg && packT && misc && film && grad_y && workspace && grad_layer && grad_misc && grad_film, "NULL argument");
    TD_REQUIRE(batch > 0 && seq_len > 0, "batch and seq_len must be positive");
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const int L = seq_len, N = cfg->layers, D = g.dirs, H = cfg->hidden, G = cfg->gru ? 3 : 4;
    const TdTrainLayout t = td_train_layout(cfg, g, batch, L);
    const TdPackTGeom tg = td_packT_geom(cfg, g);
    char *ws = (char *)workspace;
    const int DH = D * H;
    TdSideStream *side = nullptr;

    // ---- head --------------------------------------------------------------------------------
    TD_HIP(hipMemsetAsync(grad_film, 0, (size_t)batch * 2 * DH * sizeof(float), stream));
    {
        TdHeadBwdParams p;
        p.act = ws + t.act[N - 1];
        p.ln_w = misc[TD_LN_W]; p.ln_b = misc[TD_LN_B]; p.fc_w = misc[TD_FC_W]; p.film = film; p.gy = grad_y;
        p.gout = ws + t.gout[N - 1][0];
        p.chan = (float *)(ws + t.chan); p.g1 = (float *)(ws + t.g1);
        p.d_film = grad_film; p.d_ln_w = grad_misc[TD_LN_W]; p.d_ln_b = grad_misc[TD_LN_B];
        p.d_fc_w = grad_misc[TD_FC_W]; p.d_fc_b = grad_misc[TD_FC_B];
        TD_REQUIRE(p.d_ln_w && p.d_ln_b && p.d_fc_w && p.d_fc_b, "NULL gradient pointer for the head parameters");
        p.B = batch; p.L = L; p.nbb = t.nbb; p.H = H; p.hpad = g.hpad; p.kh = g.kh; p.dirs = D; p.c_out = cfg->c_out;
        const size_t ntask = (size_t)L * t.nbb;
        int blocks = (int)((ntask + 3) / 4);
        if (blocks > 2048) blocks = 2048;
        const int DHP = D * g.hpad;
        const size_t lds_max = 160 * 1024;
        auto launch = [&](auto kern, size_t shm) -> int {
            if (shm) TD_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), shm, stream, p);
            TD_LAUNCH_CHECK();
            return TD_OK;
        };
        const int cm = cfg->c_out <= 16 ? 16 : 32;
        const size_t shm = (size_t)(1 + cm) * DHP * sizeof(float);
        if (td_head_film_frag_bytes(g, batch)) {      // D * KH <= 32 fragments per position: d h on the matrix cores
            p.ffr = (const float *)(ws + t.filmfr); p.ysave = (const float *)(ws + t.ysave); p.fc_b = misc[TD_FC_B];
            p.pq = (float *)(ws + t.pq);
            rc = td_head_bwd_mfma(p, stream);
        } else if (shm <= lds_max) {
            rc = cm == 16 ? launch(td_head_bwd_dh_kernel<16, false>, shm) : launch(td_head_bwd_dh_kernel<32, false>, shm);
        } else {
            rc = cm == 16 ? launch(td_head_bwd_dh_kernel<16, true>, 0) : launch(td_head_bwd_dh_kernel<32, true>, 0);
        }
        if (rc != TD_OK) return rc;
        // d h is on its way to the top layer's sweep; the head's parameter gradients need nothing further from this stream and
        // nothing below needs them: beside the sweeps, like the weight-gradient GEMMs (joined in td_layers_backward)
        side = td_side_for(g, t.nbb, layer_done);
        hipStream_t pstream = stream;
        if (side) {
            TD_HIP(hipEventRecord(side->fork[16], stream));
            TD_HIP(hipStreamWaitEvent(side->stream2, side->fork[16], 0));
            pstream = side->stream2;
        }
        const int rows = batch * cfg->c_out;
        hipLaunchKernelGGL(td_rowsum_kernel, dim3(td_cdiv(rows, 4)), dim3(256), 0, pstream, grad_y, p.g1, rows, L);
        TD_LAUNCH_CHECK();
        // about 2048 waves per launch, chunks of at least 16 steps
        p.lch = 2048 / (t.nbb * D * g.kh);
        if (p.lch > L / 16) p.lch = L / 16;
        if (p.lch < 1) p.lch = 1;
        p.llen = td_cdiv(L, p.lch);
        p.lch = td_cdiv(L, p.llen);
        hipLaunchKernelGGL((td_head_bwd_param_kernel<16>), dim3(t.nbb, td_cdiv(D * g.kh, 4), td_cdiv(cfg->c_out, 16) * p.lch), dim3(256), 0,
                           pstream, p);
        TD_LAUNCH_CHECK();
        // the time-embedding MLP's backward (d film is complete now): with the head's parameter gradients, beside the sweeps
        if (t_steps) {
            rc = td_rnn_time_film_backward(cfg, misc, t_steps, batch, grad_film, grad_misc, film_ws, (void *)pstream);
            if (rc != TD_OK) return rc;
        }
    }
    if (layer_done) layer_done(N, user);   // LayerNorm / fc gradients are enqueued
    return td_layers_backward(cfg, g, packT, batch, L, ws, grad_layer, layer_done, user, stream, side);
}


// ---------------------------------------------------------------------------------------------
// TSTR classifier (evaluate/nn_classifier.py:89-127): RNN over (B, C, L) -> the final hidden state of each direction
// (forward: position L-1, backward: position 0) -> Linear(D*H, 1) -> Sigmoid, trained with binary cross entropy
// (nn_classifier.py:140-165).  The recurrent layers, their saves and their backward pass are the denoiser's
// (td_rnn_run_layers_ex / td_layers_backward on the training workspace); only this head is new.
// ---------------------------------------------------------------------------------------------
struct TdClsParams {
    const char *act;         // top layer's output fragments
    char *gout;              // d h_top fragments (zeroed before the backward kernel runs)
    const float *w, *b, *y;  // Linear(D*H, 1) weight, bias; labels (or NULL: gprob)
    const float *gprob;      // d loss / d prob when the loss is not this file's mean BCE
    float *prob;             // sigmoid(logit), (B)
    float *d_w, *d_b;        // accumulated
    int B, L, nbb, H, kh, dirs;
};

// one wave per batch block; lane (m, hf) holds 8 units per fragment of sequence m
template <bool BWD>
__global__ __launch_bounds__(64) void td_cls_head_kernel(TdClsParams p) {
    const int bblk = blockIdx.x, lane = threadIdx.x, m = lane & 31, hf = lane >> 5;
    const int b = bblk * 32 + m, KT = p.dirs * p.kh;
    const bool valid = b < p.B;
    float dl = 0.f;
    if (BWD && valid) {
        const float pr = p.prob[b];
        dl = p.y ? (pr - p.y[b]) / (float)p.B        // d BCE(mean) / d logit
                 : p.gprob[b] * pr * (1.0f - pr);    // through the sigmoid
    }
    float logit = 0.f;
    for (int d = 0; d < p.dirs; ++d) {
        const int l = d ? 0 : p.L - 1;                                   // where direction d's last step sits
        const size_t frag0 = ((size_t)l * p.nbb + bblk) * KT + d * p.kh;
        for (int ks = 0; ks < p.kh; ++ks) {
            const bf16x8 v = *((const bf16x8 *)(p.act + (frag0 + ks) * TD_FRAG_BYTES) + lane);
            bf16x8 g;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int unit = 16 * ks + 8 * (j >> 2) + 4 * hf + (j & 3);
                const float wv = unit < p.H ? p.w[d * p.H + unit] : 0.f;
                if (!BWD) logit = fmaf(wv, (float)v[j], logit);
                else {
                    g[j] = (__bf16)(dl * wv);
                    // d w[f] = sum over sequences of d logit * h[f]: over the 32 lanes of the half-wave, then one atomic
                    float t = dl * (float)v[j];
#pragma unroll
                    for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o);
                    if (m == 0 && unit < p.H) atomicAdd(&p.d_w[d * p.H + unit], t);
                }
            }
            if (BWD) *((bf16x8 *)(p.gout + (frag0 + ks) * TD_FRAG_BYTES) + lane) = g;
        }
    }
    if (!BWD) {
        logit += __shfl_xor(logit, 32);
        if (hf == 0 && valid) p.prob[b] = 1.0f / (1.0f + expf(-(logit + p.b[0])));
    } else {
        float t = hf == 0 ? dl : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o);
        if (lane == 0) atomicAdd(p.d_b, t);
    }
}

extern "C" size_t td_cls_workspace_bytes(const td_rnn_config *cfg, int32_t batch, int32_t seq_len) {
    return td_rnn_train_workspace_bytes(cfg, batch, seq_len);
}

// x (B, c_in, L) -> prob (B).  The workspace keeps what td_cls_backward needs.
extern "C" int td_cls_forward(const td_rnn_config *cfg, const void *packed, const float *x, const float *w, const float *b,
                              int32_t batch, int32_t seq_len, float *prob, void *workspace, void *stream_) {
    TD_REQUIRE(cfg && packed && x && w && b && prob && workspace, "NULL argument");
    TD_REQUIRE(batch > 0 && seq_len > 0, "batch and seq_len must be positive");
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    TD_REQUIRE(g.dirs * g.kh <= 64, "hidden size beyond the classifier head's range");
    hipStream_t stream = (hipStream_t)stream_;
    const TdTrainLayout t = td_train_layout(cfg, g, batch, seq_len);
    char *ws = (char *)workspace;
    char *acts[16];
    float *saves[16];
    for (int l = 0; l < cfg->layers; ++l) {
        acts[l] = ws + t.act[l];
        saves[l] = (float *)(ws + t.save[l]);
    }
    rc = td_rnn_run_layers_ex(cfg, packed, x, batch, seq_len, ws + t.x0, acts, saves, (float *)(ws + t.dc[0]), stream, nullptr, 0, -1);
    if (rc != TD_OK) return rc;
    TdClsParams p;
    p.act = acts[cfg->layers - 1]; p.gout = nullptr; p.w = w; p.b = b; p.y = nullptr; p.gprob = nullptr; p.prob = prob; p.d_w = nullptr; p.d_b = nullptr;
    p.B = batch; p.L = seq_len; p.nbb = t.nbb; p.H = cfg->hidden; p.kh = g.kh; p.dirs = g.dirs;
    hipLaunchKernelGGL((td_cls_head_kernel<false>), dim3(t.nbb), dim3(64), 0, stream, p);
    TD_LAUNCH_CHECK();
    // plain copies for the weight-gradient GEMMs of the backward pass
    hipLaunchKernelGGL(td_x_to_plain_kernel, dim3(2048), dim3(256), 0, stream, x, (__bf16 *)(ws + t.plain[0]), batch, cfg->c_in,
                       t.cpad, seq_len, t.Bp);
    TD_LAUNCH_CHECK();
    for (int l = 0; l < cfg->layers; ++l) {
        hipLaunchKernelGGL(td_frags_to_plain_kernel, dim3(td_cdiv(t.nbb, 2), seq_len, td_cdiv(g.dirs * g.kh, TD_F2P_FRAGS)), dim3(256), 0,
                           stream, (const __bf16 *)acts[l], (__bf16 *)(ws + t.plain[l + 1]), seq_len, t.nbb, g.kh, g.dirs);
        TD_LAUNCH_CHECK();
    }
    return TD_OK;
}

// Backward through the head and the recurrent layers of the last td_cls_forward on this workspace, of mean binary cross
// entropy(prob, y) when y is given, otherwise of a loss whose gradient w.r.t. prob is grad_prob: ACCUMULATES into grad_layer (4 tensors per layer and direction, torch layout), grad_w (D*H), grad_b (1).
extern "C" int td_cls_backward(const td_rnn_config *cfg, const void *packT, const float *w, const float *prob, const float *y,
                               const float *grad_prob, int32_t batch, int32_t seq_len, void *workspace, float *const *grad_layer,
                               float *grad_w, float *grad_b, void *stream_) {
    TD_REQUIRE(cfg && packT && w && prob && (y || grad_prob) && workspace && grad_layer && grad_w && grad_b, "NULL argument");
    TD_REQUIRE(batch > 0 && seq_len > 0, "batch and seq_len must be positive");
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    hipStream_t stream = (hipStream_t)stream_;
    const TdTrainLayout t = td_train_layout(cfg, g, batch, seq_len);
    char *ws = (char *)workspace;
    const int top = cfg->layers - 1;
    TD_HIP(hipMemsetAsync(ws + t.gout[top][0], 0, t.act_bytes, stream));
    TdClsParams p;
    p.act = ws + t.act[top]; p.gout = ws + t.gout[top][0]; p.w = w; p.b = nullptr; p.y = y; p.gprob = grad_prob; p.prob = (float *)prob;
    p.d_w = grad_w; p.d_b = grad_b;
    p.B = batch; p.L = seq_len; p.nbb = t.nbb; p.H = cfg->hidden; p.kh = g.kh; p.dirs = g.dirs;
    hipLaunchKernelGGL((td_cls_head_kernel<true>), dim3(t.nbb), dim3(64), 0, stream, p);
    TD_LAUNCH_CHECK();
    return td_layers_backward(cfg, g, packT, batch, seq_len, ws, grad_layer, nullptr, nullptr, stream);
}

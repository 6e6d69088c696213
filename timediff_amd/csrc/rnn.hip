// Denoiser forward: weight packing, time embedding + FiLM, layer-0 input packing, the recurrent
// layers (rnn_layer.h) and the LayerNorm + FiLM + fc head.  Reference: models/ETDiff/blocks.py.
#include "rnn_launch.h"
#include "rnn_geom.h"

#include <math.h>

extern "C" size_t td_rnn_packed_bytes(const td_rnn_config *cfg) {
    TdRnnGeom g;
    if (td_rnn_geom(cfg, &g) != TD_OK) return 0;
    return g.packed_bytes;
}

// workspace: [x0 fragments][act A][act B]
static void td_ws_layout(const TdRnnGeom &g, int batch, int L, size_t *x0_bytes, size_t *act_bytes) {
    size_t nbb = (size_t)td_cdiv(batch, TD_WAVE_BATCH);
    *x0_bytes = (size_t)L * nbb * g.g0.xfr * TD_FRAG_BYTES;
    *act_bytes = (size_t)L * nbb * (g.dirs * g.kh) * TD_FRAG_BYTES;
}

extern "C" size_t td_rnn_workspace_bytes(const td_rnn_config *cfg, int32_t batch, int32_t seq_len) {
    TdRnnGeom g;
    if (td_rnn_geom(cfg, &g) != TD_OK || batch <= 0 || seq_len <= 0) return 0;
    size_t x0, act;
    td_ws_layout(g, batch, seq_len, &x0, &act);
    return x0 + 2 * act;
}

// ---------------------------------------------------------------------------------------------
// weight packing (a derived cache of the torch-layout parameters)
// ---------------------------------------------------------------------------------------------
struct TdPackParams {
    const float *w_ih, *w_hh, *b_ih, *b_hh;
    char *dst;
    int H, In, nu, ng, kh, kx, fu, nf, ns, layer0, gru;
};

__global__ void td_pack_weights_kernel(TdPackParams p) {
    const size_t total = (size_t)p.ns * TD_STAGE_FRAGS * 512;
    __bf16 *dst = (__bf16 *)p.dst;
    for (size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (size_t)gridDim.x * blockDim.x) {
        const int f = (int)(idx / 512), lane = (int)((idx % 512) / 8), j = (int)(idx % 8);
        const int r = lane & 31, hf = lane >> 5;
        float val = 0.f;
        if (f < p.nf) {
            const int u = f / p.fu, rem0 = f % p.fu;
            const int unit = 32 * u + r;
            int g, part = 0;
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
    for (int l = 0; l < cfg->layers; ++l)
        for (int d = 0; d < cfg->dirs; ++d) {
            const TdLayerGeom &lg = l == 0 ? g.g0 : g.g1;
            const float *const *w = lw + (size_t)(l * cfg->dirs + d) * 4;
            TD_REQUIRE(w[0] && w[1] && w[2] && w[3], "NULL weight pointer for layer %d dir %d", l, d);
            TdPackParams p;
            p.w_ih = w[0]; p.w_hh = w[1]; p.b_ih = w[2]; p.b_hh = w[3];
            p.dst = (char *)packed + g.layer_off[l] + (size_t)d * lg.dir_bytes;
            p.H = cfg->hidden; p.In = l == 0 ? cfg->c_in : cfg->dirs * cfg->hidden;
            p.nu = lg.nu; p.ng = lg.ng; p.kh = lg.kh; p.kx = lg.kx; p.fu = lg.fu; p.nf = lg.nf; p.ns = lg.ns;
            p.layer0 = l == 0; p.gru = cfg->gru;
            size_t total = (size_t)lg.ns * TD_STAGE_FRAGS * 512;
            int blocks = (int)((total + 255) / 256);
            if (blocks > 4096) blocks = 4096;
            hipLaunchKernelGGL(td_pack_weights_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
            TD_LAUNCH_CHECK();
        }
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
// head: LayerNorm(D*H) -> FiLM -> fc -> (B, C_out, L)   (blocks.py:293-297, :340-341)
// one wave per (l, batch block): lane (m, hf) holds half of sequence m's features.
// ---------------------------------------------------------------------------------------------
struct TdHeadParams {
    const char *act;  // last layer's fragments
    const float *ln_w, *ln_b, *fc_w, *fc_b, *film;
    const int32_t *film_row;
    float *y;
    int film_rows, B, L, nbb, H, hpad, kh, dirs, c_out;
};

template <int CMAX>
__global__ __launch_bounds__(256) void td_head_kernel(TdHeadParams p) {
    extern __shared__ __attribute__((aligned(16))) float hl[];
    const int DHP = p.dirs * p.hpad, DH = p.dirs * p.H;
    float *s_lnw = hl, *s_lnb = hl + DHP, *s_fcw = hl + 2 * DHP;  // fc_w as [c][DHP]
    for (int i = threadIdx.x; i < DHP; i += 256) {
        const int d = i / p.hpad, unit = i % p.hpad;
        const bool ok = unit < p.H;
        s_lnw[i] = ok ? p.ln_w[d * p.H + unit] : 0.f;
        s_lnb[i] = ok ? p.ln_b[d * p.H + unit] : 0.f;
        for (int c = 0; c < CMAX; ++c) s_fcw[c * DHP + i] = (ok && c < p.c_out) ? p.fc_w[(size_t)c * DH + d * p.H + unit] : 0.f;
    }
    __syncthreads();
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, m = lane & 31, hf = lane >> 5;
    const int KT = p.dirs * p.kh;
    const size_t ntask = (size_t)p.L * p.nbb;
    for (size_t task = (size_t)blockIdx.x * 4 + wave; task < ntask; task += (size_t)gridDim.x * 4) {
        const int l = (int)(task / p.nbb), bblk = (int)(task % p.nbb);
        const int b = bblk * 32 + m;
        const bf16x8 *fp = (const bf16x8 *)(p.act + task * KT * TD_FRAG_BYTES) + lane;
        // pass 1: mean
        float sum = 0.f;
        for (int ks = 0; ks < KT; ++ks) {
            const bf16x8 v = fp[ks * 64];
#pragma unroll
            for (int j = 0; j < 8; ++j) sum += (float)v[j];  // padded units hold exact zeros
        }
        sum += __shfl_xor(sum, 32);
        const float mu = sum / (float)DH;
        float sq = 0.f;
        for (int ks = 0; ks < KT; ++ks) {
            const bf16x8 v = fp[ks * 64];
            const int unit0 = 16 * (ks % p.kh) + 4 * hf;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int unit = unit0 + 8 * (j >> 2) + (j & 3);
                const float dlt = (float)v[j] - mu;
                sq += unit < p.H ? dlt * dlt : 0.f;
            }
        }
        sq += __shfl_xor(sq, 32);
        const float rstd = rsqrtf(sq / (float)DH + 1e-5f);
        const int frow = p.film_row ? p.film_row[0] : (p.film_rows > 1 ? (b < p.B ? b : 0) : 0);
        const float *film = p.film + (size_t)frow * 2 * DH;
        float acc[CMAX];
#pragma unroll
        for (int c = 0; c < CMAX; ++c) acc[c] = 0.f;
        for (int ks = 0; ks < KT; ++ks) {
            const bf16x8 v = fp[ks * 64];
            const int d = ks / p.kh;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int unit0 = 16 * (ks % p.kh) + 8 * q + 4 * hf;  // 4 consecutive units
                const int ip = d * p.hpad + unit0;
                const f32x4 w4 = *(const f32x4 *)(s_lnw + ip), b4 = *(const f32x4 *)(s_lnb + ip);
                float yv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int unit = unit0 + e;
                    const bool ok = unit < p.H;
                    const int feat = d * p.H + (ok ? unit : 0);
                    const float scale = ok ? film[feat] : 0.f, shift = ok ? film[DH + feat] : 0.f;
                    const float xh = ((float)v[4 * q + e] - mu) * rstd;
                    yv[e] = ok ? (xh * w4[e] + b4[e]) * (scale + 1.0f) + shift : 0.f;
                }
#pragma unroll
                for (int c = 0; c < CMAX; ++c) {
                    const f32x4 f4 = *(const f32x4 *)(s_fcw + c * DHP + ip);
                    acc[c] += f4[0] * yv[0] + f4[1] * yv[1] + f4[2] * yv[2] + f4[3] * yv[3];
                }
            }
        }
#pragma unroll
        for (int c = 0; c < CMAX; ++c) acc[c] += __shfl_xor(acc[c], 32);
        if (hf == 0 && b < p.B) {
#pragma unroll
            for (int c = 0; c < CMAX; ++c)
                if (c < p.c_out) p.y[((size_t)b * p.c_out + c) * p.L + l] = acc[c] + p.fc_b[c];
        }
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
// layer dispatch (templates in rnn_launch.h; the SAVE = true instantiations live in rnn_train.hip)
// ---------------------------------------------------------------------------------------------
int td_launch_layer_infer(int gru, int nu, const TdLayerParams &p, bool layer0, int kxsel, hipStream_t stream) {
    return gru ? td_launch_layer_any<true, false>(nu, p, layer0, kxsel, stream)
               : td_launch_layer_any<false, false>(nu, p, layer0, kxsel, stream);
}
int td_launch_layer_save(int gru, int nu, const TdLayerParams &p, bool layer0, int kxsel, hipStream_t stream);

int td_profile_begin(void *stream);
void td_profile_end(int slot, void *stream);

// Runs pack-input + all recurrent layers.  acts[l] receives layer l's output fragments; saves[l] (training)
// the per-step gate activations.  Inference ping-pongs two activation buffers.
int td_rnn_run_layers_ex(const td_rnn_config *cfg, const void *packed, const float *x, int batch, int L, char *x0,
                         char *const *acts, float *const *saves, hipStream_t stream) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    const int nbb = td_cdiv(batch, TD_WAVE_BATCH);
    {
        size_t total = (size_t)L * nbb * g.kx0 * 64;
        int blocks = (int)((total + 255) / 256);
        if (blocks > 8192) blocks = 8192;
        hipLaunchKernelGGL(td_pack_input_kernel, dim3(blocks), dim3(256), 0, stream, x, x0, batch, cfg->c_in, L, nbb, g.kx0);
        TD_LAUNCH_CHECK();
    }
    const char *in = x0;
    for (int l = 0; l < cfg->layers; ++l) {
        const TdLayerGeom &lg = l == 0 ? g.g0 : g.g1;
        TdLayerParams p;
        p.wpacked = (const char *)packed + g.layer_off[l];
        p.dir_stride = lg.dir_bytes;
        p.xin = in; p.out = acts[l];
        p.in_frags = lg.xfr; p.out_frags = g.dirs * g.kh;
        p.L = L; p.nbb = nbb; p.dirs = g.dirs;
        p.save = saves ? saves[l] : nullptr;
        p.save_dir_stride = (size_t)L * nbb * g.nu * 5 * 1024;
        const int kxsel = l == 0 ? g.kx0 : g.dirs;
        if (saves) {
            rc = td_launch_layer_save(cfg->gru, g.nu, p, l == 0, kxsel, stream);
        } else {
            const int slot = l > 0 ? td_profile_begin(stream) : -1;
            rc = td_launch_layer_infer(cfg->gru, g.nu, p, l == 0, kxsel, stream);
            td_profile_end(slot, stream);
        }
        if (rc != TD_OK) return rc;
        in = acts[l];
    }
    return TD_OK;
}

int td_rnn_run_layers(const td_rnn_config *cfg, const void *packed, const float *x, int batch, int L, void *workspace,
                      hipStream_t stream, const char **last_act) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    size_t x0_bytes, act_bytes;
    td_ws_layout(g, batch, L, &x0_bytes, &act_bytes);
    char *x0 = (char *)workspace, *actA = x0 + x0_bytes, *actB = actA + act_bytes;
    char *acts[16];
    for (int l = 0; l < cfg->layers; ++l) acts[l] = (l & 1) ? actB : actA;
    *last_act = acts[cfg->layers - 1];
    return td_rnn_run_layers_ex(cfg, packed, x, batch, L, x0, acts, nullptr, stream);
}

int td_rnn_head(const td_rnn_config *cfg, const float *const *misc, const char *act, const float *film, int film_rows,
                const int32_t *film_row, int batch, int L, float *y, hipStream_t stream) {
    TdRnnGeom g;
    int rc = td_rnn_geom(cfg, &g);
    if (rc != TD_OK) return rc;
    TdHeadParams p;
    p.act = act;
    p.ln_w = misc[TD_LN_W]; p.ln_b = misc[TD_LN_B]; p.fc_w = misc[TD_FC_W]; p.fc_b = misc[TD_FC_B];
    p.film = film; p.film_row = film_row; p.film_rows = film_rows;
    p.y = y; p.B = batch; p.L = L; p.nbb = td_cdiv(batch, TD_WAVE_BATCH);
    p.H = cfg->hidden; p.hpad = g.hpad; p.kh = g.kh; p.dirs = g.dirs; p.c_out = cfg->c_out;
    const size_t ntask = (size_t)L * p.nbb;
    int blocks = (int)((ntask + 3) / 4);
    if (blocks > 2048) blocks = 2048;
    const int DHP = g.dirs * g.hpad;
    if (cfg->c_out <= 16) {
        size_t shm = (size_t)(2 + 16) * DHP * sizeof(float);
        hipLaunchKernelGGL(td_head_kernel<16>, dim3(blocks), dim3(256), shm, stream, p);
    } else {
        size_t shm = (size_t)(2 + 32) * DHP * sizeof(float);
        TD_HIP(hipFuncSetAttribute((const void *)td_head_kernel<32>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm));
        hipLaunchKernelGGL(td_head_kernel<32>, dim3(blocks), dim3(256), shm, stream, p);
    }
    TD_LAUNCH_CHECK();
    return TD_OK;
}

extern "C" int td_rnn_forward(const td_rnn_config *cfg, const void *packed, const float *const *misc, const float *x,
                              const float *film, int32_t film_rows, const int32_t *film_row, int32_t batch,
                              int32_t seq_len, float *y, void *workspace, void *stream) {
    TD_REQUIRE(cfg && packed && misc && x && film && y && workspace, "NULL argument");
    TD_REQUIRE(batch > 0 && seq_len > 0, "batch and seq_len must be positive");
    TD_REQUIRE(film_rows == 1 || film_rows == batch || film_row, "film_rows must be 1 or batch");
    const char *act = nullptr;
    int rc = td_rnn_run_layers(cfg, packed, x, batch, seq_len, workspace, (hipStream_t)stream, &act);
    if (rc != TD_OK) return rc;
    return td_rnn_head(cfg, misc, act, film, film_rows, film_row, batch, seq_len, y, (hipStream_t)stream);
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

/*
 * etdiff_oracle.c -- CPU restatement of the ETDiff hot path.   TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library,
 * and only as the checker / the timed CPU baseline.  The product (timediff_amd/) never links,
 * imports or falls back to it.
 *
 * Parity pin: every function below is checked against golden vectors produced by running the
 * reference's own Python modules (tests/golden/make_golden.py writes tests/golden/ npz files,
 * tests/test_oracle_golden.py).  File:line citations are relative to /root/reference.
 *
 * Layouts are torch's: tensors row-major, LSTM gate rows [i,f,g,o], GRU gate rows [r,z,n].
 * Arithmetic is fp32 like the reference; `bf16` switches on an emulation of the HIP denoiser's
 * operand rounding (bf16 GEMM operands, fp32 accumulate, fp32 cell state) so the MFMA kernel can be
 * compared tightly as well as against the plain fp32 restatement.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

/* ------------------------------------------------------------------------------------------ */
/* small helpers                                                                              */
/* ------------------------------------------------------------------------------------------ */
static inline float bf16_round(float x) {
    union { float f; uint32_t u; } v;
    v.f = x;
    if ((v.u & 0x7fffffffu) > 0x7f800000u) return x; /* NaN stays */
    v.u = (v.u + 0x7fffu + ((v.u >> 16) & 1u)) & 0xffff0000u;
    return v.f;
}
static inline float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

static inline float dotf(const float *a, const float *b, int n) {
    float acc = 0.f;
#pragma omp simd reduction(+ : acc)
    for (int i = 0; i < n; ++i) acc += a[i] * b[i];
    return acc;
}

/* mixed_diffusion.py:20-22 */
static inline float log_add_exp(float a, float b) {
    float m = a > b ? a : b;
    return m + logf(expf(a - m) + expf(b - m));
}
/* mixed_diffusion.py:27-30 */
static inline float log_sub_exp(float a, float b) {
    float m = a > b ? a : b;
    return logf(expf(a - m) - expf(b - m)) + m;
}

#define LOG_1E30 (-69.07755279f) /* log(clamp(onehot, 1e-30)), mixed_diffusion.py:50 */

/* ------------------------------------------------------------------------------------------ */
/* counter-based RNG shared with the HIP kernels (Philox4x32-10).  No counterpart in the     */
/* reference (it draws from torch's global generator); parity runs feed external noise.       */
/* ------------------------------------------------------------------------------------------ */
static inline void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                                 uint32_t out[4]) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
static inline float u32_to_unit(uint32_t x) { return (float)(x >> 8) * (1.0f / 16777216.0f); }          /* [0,1) */
static inline float u32_to_unit_open(uint32_t x) { return ((float)(x >> 8) + 1.0f) * (1.0f / 16777216.0f); } /* (0,1] */

/* Draw layout (identical in timediff_amd/csrc/rng.h): counter = (patient id, position l, tag, block).
 * Normal block j (c3 = j) yields channels 4j..4j+3 by two Box-Muller pairs; uniform block j
 * (c3 = 0x10000 + j) yields the flat class slots 5j..5j+4 (five 24-bit fields of the 128 output bits). */
EXPORT void or_philox_normals(uint64_t seed, uint32_t patient, uint32_t l, uint32_t tag, int n, float *out) {
    for (int j = 0; j * 4 < n; ++j) {
        uint32_t r[4];
        philox4x32_10(patient, l, tag, (uint32_t)j, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        float z[4];
        for (int p = 0; p < 2; ++p) {
            float u1 = u32_to_unit_open(r[2 * p]), u2 = u32_to_unit(r[2 * p + 1]);
            float rad = sqrtf(-2.0f * logf(u1));
            double ang = 6.283185307179586476925 * (double)u2;   /* exact angle: the kernels use sincospi(2 u2) */
            z[2 * p] = rad * (float)cos(ang);
            z[2 * p + 1] = rad * (float)sin(ang);
        }
        for (int q = 0; q < 4 && j * 4 + q < n; ++q) out[j * 4 + q] = z[q];
    }
}
EXPORT void or_philox_uniforms(uint64_t seed, uint32_t patient, uint32_t l, uint32_t tag, int n, float *out) {
    /* five 24-bit fields per 128-bit block, most significant first (timediff_amd/csrc/rng.h td_uniform5) */
    for (int j = 0; j * 5 < n; ++j) {
        uint32_t r[4], f[5];
        philox4x32_10(patient, l, tag, 0x10000u + (uint32_t)j, (uint32_t)seed, (uint32_t)(seed >> 32), r);
        f[0] = r[0] >> 8;
        f[1] = ((r[0] & 0xFFu) << 16) | (r[1] >> 16);
        f[2] = ((r[1] & 0xFFFFu) << 8) | (r[2] >> 24);
        f[3] = r[2] & 0xFFFFFFu;
        f[4] = r[3] >> 8;
        for (int q = 0; q < 5 && j * 5 + q < n; ++q) out[j * 5 + q] = (float)f[q] * (1.0f / 16777216.0f);
    }
}

/* ------------------------------------------------------------------------------------------ */
/* sliced_logsumexp, reference-faithful (mixed_diffusion.py:32-42): differences of a cumulative */
/* logcumsumexp over the padded class axis.  torch's CPU logcumsumexp accumulates in double.   */
/* x: (B, S, L), offsets: (n+1).  out: (B, S, L) with each slice's LSE repeated over the slice. */
/* ------------------------------------------------------------------------------------------ */
static void sliced_lse_faithful_col(const float *x, int stride, const int64_t *offsets, int n_slices, float *lse_out) {
    /* one (b,l) column: x[k*stride], k < S */
    int S = (int)offsets[n_slices];
    float cum[S + 1];
    double acc = -INFINITY;
    cum[0] = -INFINITY;
    for (int k = 0; k < S; ++k) {
        double v = (double)x[(size_t)k * stride];
        double mn = acc < v ? acc : v, mx = acc < v ? v : acc;
        if (mn != mx || isfinite(mn)) acc = log1p(exp(mn - mx)) + mx;
        cum[k + 1] = (float)acc;
    }
    for (int s = 0; s < n_slices; ++s) lse_out[s] = log_sub_exp(cum[offsets[s + 1]], cum[offsets[s]]);
}

EXPORT void or_sliced_logsumexp(const float *x, int B, int S, int L, const int64_t *offsets, int n_slices, float *out) {
    for (int b = 0; b < B; ++b)
        for (int l = 0; l < L; ++l) {
            float lse[n_slices];
            sliced_lse_faithful_col(x + ((size_t)b * S) * L + l, L, offsets, n_slices, lse);
            for (int s = 0; s < n_slices; ++s)
                for (int k = (int)offsets[s]; k < (int)offsets[s + 1]; ++k) out[((size_t)b * S + k) * L + l] = lse[s];
        }
}

/* ------------------------------------------------------------------------------------------ */
/* Denoiser: RNN.forward (blocks.py:333-341) = time_emb (:324-329, :17-24) + RNNBlock (:282-297) */
/* + fc.                                                                                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t C_in, H, C_out, N, D, gru, embed_dim, time_dim, bf16;
} OrRnnCfg;

/* weights: lw[(l*D+d)*4 + {0:w_ih,1:w_hh,2:b_ih,3:b_hh}]; misc[0..9] = te_w1, te_b1, te_w2, te_b2,
 * film_w, film_b, ln_w, ln_b, fc_w, fc_b. */
enum { M_TE_W1, M_TE_B1, M_TE_W2, M_TE_B2, M_FILM_W, M_FILM_B, M_LN_W, M_LN_B, M_FC_W, M_FC_B, M_COUNT };

static void time_embed(const OrRnnCfg *c, const float *const *misc, int64_t t, float *sinus, float *h1pre, float *h1,
                       float *temb) {
    int half = c->embed_dim / 2;
    double emb = log(10000.0) / (half - 1);
    for (int j = 0; j < half; ++j) {
        float f = expf((float)j * (float)(-emb));
        float a = (float)t * f;
        sinus[j] = sinf(a);
        sinus[half + j] = cosf(a);
    }
    for (int r = 0; r < c->time_dim; ++r) {
        float v = dotf(misc[M_TE_W1] + (size_t)r * c->embed_dim, sinus, c->embed_dim) + misc[M_TE_B1][r];
        h1pre[r] = v;
        h1[r] = 0.5f * v * (1.0f + erff(v * 0.70710678118654752f)); /* nn.GELU() exact */
    }
    for (int r = 0; r < c->time_dim; ++r)
        temb[r] = dotf(misc[M_TE_W2] + (size_t)r * c->time_dim, h1, c->time_dim) + misc[M_TE_B2][r];
}

EXPORT void or_time_embed(const OrRnnCfg *c, const float *const *misc, const int64_t *t, int B, float *temb_out,
                          float *film_out) {
    int DH = c->D * c->H;
    for (int b = 0; b < B; ++b) {
        float sinus[c->embed_dim], h1pre[c->time_dim], h1[c->time_dim], temb[c->time_dim], s[c->time_dim];
        time_embed(c, misc, t[b], sinus, h1pre, h1, temb);
        if (temb_out) memcpy(temb_out + (size_t)b * c->time_dim, temb, sizeof(float) * c->time_dim);
        if (film_out) {
            for (int k = 0; k < c->time_dim; ++k) s[k] = temb[k] * sigmoidf_(temb[k]);
            for (int r = 0; r < 2 * DH; ++r)
                film_out[(size_t)b * 2 * DH + r] =
                    dotf(misc[M_FILM_W] + (size_t)r * c->time_dim, s, c->time_dim) + misc[M_FILM_B][r];
        }
    }
}

/* One direction of one recurrent layer for one sequence.
 * in: (L, In) row-major; out: written into out[(l)*out_stride + out_off + j].
 * If act != NULL it receives, per time step, G*H activated gate values then (LSTM) H cell values:
 * LSTM: [i,f,g,o,c]  (5H);  GRU: [r,z,n,hn] (4H) with hn = W_hn h + b_hn.  */
static void rnn_dir(const OrRnnCfg *c, const float *w_ih, const float *w_hh, const float *b_ih, const float *b_hh,
                    const float *in, int In, int L, int reverse, int layer, float *out, int out_stride, int out_off,
                    float *act) {
    const int H = c->H, G = c->gru ? 3 : 4;
    const int emul = c->bf16;
    float *wih_r = NULL, *whh_r = NULL;
    if (emul) {
        /* layer 0's input projection stays ~fp32 on the GPU (split-bf16 x3), everything else bf16 */
        /* the GPU rounds the weights after folding in the factor its exp2-based gate math needs:
         * -log2(e) for the sigmoid gates, 2 log2(e) for the tanh gate (LSTM g, GRU n) */
        whh_r = (float *)malloc(sizeof(float) * G * H * H);
        for (int i = 0; i < G * H * H; ++i) {
            const float sc = (i / (H * H)) == 2 ? 2.8853900817779268f : -1.4426950408889634f;
            whh_r[i] = bf16_round(w_hh[i] * sc) / sc;
        }
        w_hh = whh_r;
        if (layer > 0) {
            wih_r = (float *)malloc(sizeof(float) * G * H * In);
            for (int i = 0; i < G * H * In; ++i) {
                const float sc = (i / (H * In)) == 2 ? 2.8853900817779268f : -1.4426950408889634f;
                wih_r[i] = bf16_round(w_ih[i] * sc) / sc;
            }
            w_ih = wih_r;
        }
    }
    float h[H], cst[H], hq[H], pre[4 * H], hn[H];
    memset(h, 0, sizeof h);
    memset(cst, 0, sizeof cst);
    memset(hq, 0, sizeof hq);
    for (int s = 0; s < L; ++s) {
        int l = reverse ? L - 1 - s : s;
        const float *x = in + (size_t)l * In;
        if (!c->gru) {
            for (int r = 0; r < 4 * H; ++r)
                pre[r] = dotf(w_ih + (size_t)r * In, x, In) + dotf(w_hh + (size_t)r * H, hq, H) + (b_ih[r] + b_hh[r]);
            for (int j = 0; j < H; ++j) {
                float ig = sigmoidf_(pre[j]), fg = sigmoidf_(pre[H + j]), gg = tanhf(pre[2 * H + j]),
                      og = sigmoidf_(pre[3 * H + j]);
                cst[j] = fg * cst[j] + ig * gg;
                h[j] = og * tanhf(cst[j]);
                if (act) {
                    float *a = act + (size_t)l * 5 * H;
                    a[j] = ig; a[H + j] = fg; a[2 * H + j] = gg; a[3 * H + j] = og; a[4 * H + j] = cst[j];
                }
            }
        } else {
            for (int r = 0; r < 3 * H; ++r) {
                float xi = dotf(w_ih + (size_t)r * In, x, In) + b_ih[r];
                float hh = dotf(w_hh + (size_t)r * H, hq, H) + b_hh[r];
                if (r < 2 * H) pre[r] = xi + hh;
                else { pre[r] = xi; hn[r - 2 * H] = hh; }
            }
            for (int j = 0; j < H; ++j) {
                float rg = sigmoidf_(pre[j]), zg = sigmoidf_(pre[H + j]);
                float ng = tanhf(pre[2 * H + j] + rg * hn[j]);
                /* the cell blends with the un-rounded previous state, as the kernel keeps it in fp32 */
                h[j] = (1.0f - zg) * ng + zg * h[j];
                if (act) {
                    float *a = act + (size_t)l * 4 * H;
                    a[j] = rg; a[H + j] = zg; a[2 * H + j] = ng; a[3 * H + j] = hn[j];
                }
            }
        }
        for (int j = 0; j < H; ++j) {
            float hv = emul ? bf16_round(h[j]) : h[j];
            hq[j] = hv;
            out[(size_t)l * out_stride + out_off + j] = hv;
        }
    }
    free(wih_r);
    free(whh_r);
}

/* Inter-layer dropout (nn.LSTM / nn.GRU `dropout=`, blocks.py:268-270) with the masks GIVEN: (N-1, B, L, D*H) fp32,
 * entries 0 or 1/(1-p), multiplied onto every layer's output but the top one's on its way to the next layer.  NULL: none.
 * (torch draws its masks from its own generator; a parity run hands both sides the same ones.) */
static const float *g_drop = NULL;
static int g_drop_B = 0;
EXPORT void or_set_dropout_masks(const float *masks, int B) { g_drop = masks; g_drop_B = B; }
static const float *drop_mask(const OrRnnCfg *c, int layer, int b, int L) {
    return g_drop + ((size_t)layer * g_drop_B + b) * L * c->D * c->H;
}

/* Per-sequence forward.  bufs: layer outputs (N+1 of them; [0] is the transposed input). */
typedef struct {
    float **layer_out; /* [N+1] each (L, width) */
    float **layer_in;  /* [N] layer l's input: layer_out[l], or its dropped copy */
    float **act;       /* [N*D] each (L, 5H|4H) or NULL */
    float *xhat, *rstd; /* (L, DH), (L) */
    float *temb, *h1pre, *h1, *sinus, *film;
} SeqSave;

static void rnn_forward_seq(const OrRnnCfg *c, const float *const *lw, const float *const *misc, const float *x /*(C,L)*/,
                            int b, int L, int64_t t, float *y /*(C_out,L)*/, float *hidden /*(L,DH) or NULL*/, SeqSave *sv) {
    const int H = c->H, D = c->D, DH = D * H;
    float *bufA = (float *)malloc(sizeof(float) * (size_t)L * (c->C_in > DH ? c->C_in : DH));
    float *bufB = (float *)malloc(sizeof(float) * (size_t)L * DH);
    float sinus[c->embed_dim], h1pre[c->time_dim], h1[c->time_dim], temb[c->time_dim], s[c->time_dim], film[2 * DH];
    time_embed(c, misc, t, sinus, h1pre, h1, temb);
    for (int k = 0; k < c->time_dim; ++k) s[k] = temb[k] * sigmoidf_(temb[k]);
    for (int r = 0; r < 2 * DH; ++r) film[r] = dotf(misc[M_FILM_W] + (size_t)r * c->time_dim, s, c->time_dim) + misc[M_FILM_B][r];
    if (sv) {
        memcpy(sv->sinus, sinus, sizeof sinus); memcpy(sv->h1pre, h1pre, sizeof h1pre);
        memcpy(sv->h1, h1, sizeof h1); memcpy(sv->temb, temb, sizeof temb); memcpy(sv->film, film, sizeof film);
    }
    float *in = sv ? sv->layer_out[0] : bufA;
    for (int l = 0; l < L; ++l)
        for (int ch = 0; ch < c->C_in; ++ch) in[(size_t)l * c->C_in + ch] = x[(size_t)ch * L + l];
    int In = c->C_in;
    float *out = NULL;
    for (int layer = 0; layer < c->N; ++layer) {
        out = sv ? sv->layer_out[layer + 1] : (in == bufA ? bufB : bufA);
        for (int d = 0; d < D; ++d) {
            const float *const *w = lw + (size_t)(layer * D + d) * 4;
            rnn_dir(c, w[0], w[1], w[2], w[3], in, In, L, d == 1, layer, out, DH, d * H, sv ? sv->act[layer * D + d] : NULL);
        }
        in = out;
        In = DH;
        if (g_drop && layer + 1 < c->N) {
            const float *m = drop_mask(c, layer, b, L);
            float *dst = sv ? sv->layer_in[layer + 1] : out;
            for (size_t i = 0; i < (size_t)L * DH; ++i) dst[i] = out[i] * m[i];
            in = dst;
        }
    }
    if (hidden) memcpy(hidden, out, sizeof(float) * (size_t)L * DH);
    const float *ln_w = misc[M_LN_W], *ln_b = misc[M_LN_B], *fc_w = misc[M_FC_W], *fc_b = misc[M_FC_B];
    float yf[DH];
    for (int l = 0; l < L; ++l) {
        const float *hrow = out + (size_t)l * DH;
        float mu = 0.f, var = 0.f;
        for (int j = 0; j < DH; ++j) mu += hrow[j];
        mu /= DH;
        for (int j = 0; j < DH; ++j) var += (hrow[j] - mu) * (hrow[j] - mu);
        var /= DH;
        float rstd = 1.0f / sqrtf(var + 1e-5f);
        for (int j = 0; j < DH; ++j) {
            float xh = (hrow[j] - mu) * rstd;
            if (sv) sv->xhat[(size_t)l * DH + j] = xh;
            yf[j] = (xh * ln_w[j] + ln_b[j]) * (film[j] + 1.0f) + film[DH + j];
        }
        if (sv) sv->rstd[l] = rstd;
        for (int co = 0; co < c->C_out; ++co) y[(size_t)co * L + l] = dotf(fc_w + (size_t)co * DH, yf, DH) + fc_b[co];
    }
    free(bufA);
    free(bufB);
}

/* x: (B, C_in, L); t: (B); y: (B, C_out, L); hidden: (B, L, D*H) or NULL */
EXPORT void or_rnn_forward(const OrRnnCfg *c, const float *const *lw, const float *const *misc, const float *x,
                           const int64_t *t, int B, int L, float *y, float *hidden) {
#pragma omp parallel for schedule(dynamic, 1)
    for (int b = 0; b < B; ++b)
        rnn_forward_seq(c, lw, misc, x + (size_t)b * c->C_in * L, b, L, t[b], y + (size_t)b * c->C_out * L,
                        hidden ? hidden + (size_t)b * L * c->D * c->H : NULL, NULL);
}

/* Backward of the denoiser (autograd of blocks.py:333-341), fp32.  grads use the same index
 * scheme as the weights (glw / gmisc), must be zero-initialised by the caller and are
 * accumulated over the batch.  gx: (B, C_in, L) or NULL. */
static void rnn_backward_range(const OrRnnCfg *c, const float *const *lw, const float *const *misc, const float *x,
                               const int64_t *t, int b0, int b1, int L, const float *gy, float *const *glw,
                               float *const *gmisc, float *gx, float *y_out) {
    const int H = c->H, D = c->D, DH = D * H, G = c->gru ? 3 : 4, AW = c->gru ? 4 * H : 5 * H;
    for (int b = b0; b < b1; ++b) {
        SeqSave sv;
        sv.layer_out = (float **)malloc(sizeof(float *) * (c->N + 1));
        sv.act = (float **)malloc(sizeof(float *) * c->N * D);
        sv.layer_out[0] = (float *)malloc(sizeof(float) * (size_t)L * c->C_in);
        for (int l = 1; l <= c->N; ++l) sv.layer_out[l] = (float *)malloc(sizeof(float) * (size_t)L * DH);
        sv.layer_in = (float **)malloc(sizeof(float *) * c->N);
        for (int l = 0; l < c->N; ++l)
            sv.layer_in[l] = (g_drop && l > 0) ? (float *)malloc(sizeof(float) * (size_t)L * DH) : sv.layer_out[l];
        for (int i = 0; i < c->N * D; ++i) sv.act[i] = (float *)malloc(sizeof(float) * (size_t)L * AW);
        sv.xhat = (float *)malloc(sizeof(float) * (size_t)L * DH);
        sv.rstd = (float *)malloc(sizeof(float) * L);
        sv.temb = (float *)malloc(sizeof(float) * c->time_dim);
        sv.h1pre = (float *)malloc(sizeof(float) * c->time_dim);
        sv.h1 = (float *)malloc(sizeof(float) * c->time_dim);
        sv.sinus = (float *)malloc(sizeof(float) * c->embed_dim);
        sv.film = (float *)malloc(sizeof(float) * 2 * DH);
        float *y = (float *)malloc(sizeof(float) * (size_t)c->C_out * L);
        rnn_forward_seq(c, lw, misc, x + (size_t)b * c->C_in * L, b, L, t[b], y, NULL, &sv);
        if (y_out) memcpy(y_out + (size_t)b * c->C_out * L, y, sizeof(float) * (size_t)c->C_out * L);
        const float *g = gy + (size_t)b * c->C_out * L;
        const float *ln_w = misc[M_LN_W], *ln_b = misc[M_LN_B], *fc_w = misc[M_FC_W];
        /* head: fc, FiLM, LayerNorm */
        float *dh_top = (float *)calloc((size_t)L * DH, sizeof(float));
        float dfilm[2 * DH];
        memset(dfilm, 0, sizeof dfilm);
        for (int l = 0; l < L; ++l) {
            float dyf[DH], dxh[DH];
            memset(dyf, 0, sizeof dyf);
            for (int co = 0; co < c->C_out; ++co) {
                float gv = g[(size_t)co * L + l];
                gmisc[M_FC_B][co] += gv;
                for (int j = 0; j < DH; ++j) {
                    float xh = sv.xhat[(size_t)l * DH + j];
                    float yfj = (xh * ln_w[j] + ln_b[j]) * (sv.film[j] + 1.0f) + sv.film[DH + j];
                    gmisc[M_FC_W][(size_t)co * DH + j] += gv * yfj;
                    dyf[j] += gv * fc_w[(size_t)co * DH + j];
                }
            }
            float m1 = 0.f, m2 = 0.f;
            for (int j = 0; j < DH; ++j) {
                float xh = sv.xhat[(size_t)l * DH + j];
                float lnout = xh * ln_w[j] + ln_b[j];
                dfilm[j] += dyf[j] * lnout;
                dfilm[DH + j] += dyf[j];
                float dln = dyf[j] * (sv.film[j] + 1.0f);
                gmisc[M_LN_W][j] += dln * xh;
                gmisc[M_LN_B][j] += dln;
                dxh[j] = dln * ln_w[j];
                m1 += dxh[j];
                m2 += dxh[j] * xh;
            }
            m1 /= DH; m2 /= DH;
            for (int j = 0; j < DH; ++j)
                dh_top[(size_t)l * DH + j] = sv.rstd[l] * (dxh[j] - m1 - sv.xhat[(size_t)l * DH + j] * m2);
        }
        /* FiLM linear + time embedding MLP */
        {
            int TD = c->time_dim, ED = c->embed_dim;
            float s[TD], dtemb[TD], dh1[TD];
            memset(dtemb, 0, sizeof dtemb);
            for (int k = 0; k < TD; ++k) s[k] = sv.temb[k] * sigmoidf_(sv.temb[k]);
            for (int r = 0; r < 2 * DH; ++r) {
                gmisc[M_FILM_B][r] += dfilm[r];
                for (int k = 0; k < TD; ++k) {
                    gmisc[M_FILM_W][(size_t)r * TD + k] += dfilm[r] * s[k];
                    dtemb[k] += dfilm[r] * misc[M_FILM_W][(size_t)r * TD + k];
                }
            }
            for (int k = 0; k < TD; ++k) {
                float sg = sigmoidf_(sv.temb[k]);
                dtemb[k] *= sg * (1.0f + sv.temb[k] * (1.0f - sg));
            }
            memset(dh1, 0, sizeof dh1);
            for (int r = 0; r < TD; ++r) {
                gmisc[M_TE_B2][r] += dtemb[r];
                for (int k = 0; k < TD; ++k) {
                    gmisc[M_TE_W2][(size_t)r * TD + k] += dtemb[r] * sv.h1[k];
                    dh1[k] += dtemb[r] * misc[M_TE_W2][(size_t)r * TD + k];
                }
            }
            for (int r = 0; r < TD; ++r) {
                float v = sv.h1pre[r];
                float cdf = 0.5f * (1.0f + erff(v * 0.70710678118654752f));
                float pdf = 0.3989422804014327f * expf(-0.5f * v * v);
                float dpre = dh1[r] * (cdf + v * pdf);
                gmisc[M_TE_B1][r] += dpre;
                for (int k = 0; k < ED; ++k) gmisc[M_TE_W1][(size_t)r * ED + k] += dpre * sv.sinus[k];
            }
        }
        /* BPTT, top layer first */
        float *dout = dh_top; /* (L, DH) gradient w.r.t. this layer's output */
        for (int layer = c->N - 1; layer >= 0; --layer) {
            int In = layer == 0 ? c->C_in : DH;
            const float *in = sv.layer_in[layer];
            const float *outp = sv.layer_out[layer + 1];
            float *din = (float *)calloc((size_t)L * In, sizeof(float));
            for (int d = 0; d < D; ++d) {
                const float *const *w = lw + (size_t)(layer * D + d) * 4;
                float *const *gw = glw + (size_t)(layer * D + d) * 4;
                const float *act = sv.act[layer * D + d];
                float dh[H], dc[H], da[4 * H], dah[4 * H], dhp[H];
                memset(dh, 0, sizeof dh);
                memset(dc, 0, sizeof dc);
                for (int s = L - 1; s >= 0; --s) {
                    int l = d ? L - 1 - s : s;
                    int lp = d ? l + 1 : l - 1; /* previous step in this direction's order */
                    int has_prev = s > 0;
                    const float *a = act + (size_t)l * AW;
                    for (int j = 0; j < H; ++j) dh[j] += dout[(size_t)l * DH + d * H + j];
                    if (!c->gru) {
                        for (int j = 0; j < H; ++j) {
                            float ig = a[j], fg = a[H + j], gg = a[2 * H + j], og = a[3 * H + j], cv = a[4 * H + j];
                            float cprev = has_prev ? act[(size_t)lp * AW + 4 * H + j] : 0.f;
                            float tc = tanhf(cv);
                            float dcv = dc[j] + dh[j] * og * (1.0f - tc * tc);
                            da[j] = dcv * gg * ig * (1.0f - ig);
                            da[H + j] = dcv * cprev * fg * (1.0f - fg);
                            da[2 * H + j] = dcv * ig * (1.0f - gg * gg);
                            da[3 * H + j] = dh[j] * tc * og * (1.0f - og);
                            dc[j] = dcv * fg;
                        }
                        memcpy(dah, da, sizeof(float) * 4 * H);
                        memset(dhp, 0, sizeof dhp);
                    } else {
                        for (int j = 0; j < H; ++j) {
                            float rg = a[j], zg = a[H + j], ng = a[2 * H + j], hnv = a[3 * H + j];
                            float hprev = has_prev ? outp[(size_t)lp * DH + d * H + j] : 0.f;
                            float dn = dh[j] * (1.0f - zg);
                            float dz = dh[j] * (hprev - ng);
                            float dan = dn * (1.0f - ng * ng);
                            da[j] = dan * hnv * rg * (1.0f - rg);
                            da[H + j] = dz * zg * (1.0f - zg);
                            da[2 * H + j] = dan;
                            dah[j] = da[j]; dah[H + j] = da[H + j]; dah[2 * H + j] = dan * rg;
                            dhp[j] = dh[j] * zg;
                        }
                    }
                    for (int r = 0; r < G * H; ++r) {
                        gw[2][r] += da[r];
                        gw[3][r] += dah[r];
                        float *gwi = gw[0] + (size_t)r * In;
                        const float *wi = w[0] + (size_t)r * In;
                        for (int k = 0; k < In; ++k) {
                            gwi[k] += da[r] * in[(size_t)l * In + k];
                            din[(size_t)l * In + k] += da[r] * wi[k];
                        }
                        if (has_prev) {
                            float *gwh = gw[1] + (size_t)r * H;
                            for (int k = 0; k < H; ++k) gwh[k] += dah[r] * outp[(size_t)lp * DH + d * H + k];
                        }
                        const float *wh = w[1] + (size_t)r * H;
                        for (int k = 0; k < H; ++k) dhp[k] += dah[r] * wh[k];
                    }
                    memcpy(dh, dhp, sizeof dhp);
                }
            }
            if (g_drop && layer > 0) {   /* d(dropped output) -> d(output of the layer below) */
                const float *m = drop_mask(c, layer - 1, b, L);
                for (size_t i = 0; i < (size_t)L * DH; ++i) din[i] *= m[i];
            }
            free(dout);
            dout = din;
        }
        if (gx)
            for (int l = 0; l < L; ++l)
                for (int ch = 0; ch < c->C_in; ++ch) gx[((size_t)b * c->C_in + ch) * L + l] = dout[(size_t)l * c->C_in + ch];
        free(dout);
        for (int l = 1; l < c->N; ++l)
            if (sv.layer_in[l] != sv.layer_out[l]) free(sv.layer_in[l]);
        for (int l = 0; l <= c->N; ++l) free(sv.layer_out[l]);
        for (int i = 0; i < c->N * D; ++i) free(sv.act[i]);
        free(sv.layer_in); free(sv.layer_out); free(sv.act); free(sv.xhat); free(sv.rstd); free(sv.temb); free(sv.h1pre); free(sv.h1);
        free(sv.sinus); free(sv.film); free(y);
    }
}

/* The batch is cut into a FIXED number of chunks (independent of the thread count), each chunk accumulates into
 * gradient buffers of its own and the chunks are added up in order: the result does not depend on how many threads
 * ran.  (Test infrastructure: the L = 272 parity tests would otherwise wait minutes for a serial sweep.) */
#define OR_BWD_CHUNKS 16
EXPORT void or_rnn_backward(const OrRnnCfg *c, const float *const *lw, const float *const *misc, const float *x,
                            const int64_t *t, int B, int L, const float *gy, float *const *glw, float *const *gmisc,
                            float *gx, float *y_out) {
    const int H = c->H, D = c->D, DH = D * H, G = c->gru ? 3 : 4, TD = c->time_dim, ED = c->embed_dim;
    const int nch = B < OR_BWD_CHUNKS ? B : OR_BWD_CHUNKS;
    if (nch <= 1) {
        rnn_backward_range(c, lw, misc, x, t, 0, B, L, gy, glw, gmisc, gx, y_out);
        return;
    }
    const int nlw = c->N * D * 4;
    size_t lsz[nlw], msz[M_COUNT];
    for (int l = 0; l < c->N; ++l)
        for (int d = 0; d < D; ++d) {
            const int In = l == 0 ? c->C_in : DH;
            size_t *z = lsz + (size_t)(l * D + d) * 4;
            z[0] = (size_t)G * H * In; z[1] = (size_t)G * H * H; z[2] = z[3] = (size_t)G * H;
        }
    msz[M_TE_W1] = (size_t)TD * ED; msz[M_TE_B1] = TD; msz[M_TE_W2] = (size_t)TD * TD; msz[M_TE_B2] = TD;
    msz[M_FILM_W] = (size_t)2 * DH * TD; msz[M_FILM_B] = 2 * DH; msz[M_LN_W] = DH; msz[M_LN_B] = DH;
    msz[M_FC_W] = (size_t)c->C_out * DH; msz[M_FC_B] = c->C_out;
    size_t total = 0;
    for (int i = 0; i < nlw; ++i) total += lsz[i];
    for (int i = 0; i < M_COUNT; ++i) total += msz[i];
    float *pool = (float *)calloc(total * nch, sizeof(float));
#pragma omp parallel for schedule(dynamic, 1)
    for (int ch = 0; ch < nch; ++ch) {
        float *pl[nlw], *pm[M_COUNT];
        float *q = pool + total * ch;
        for (int i = 0; i < nlw; ++i) { pl[i] = q; q += lsz[i]; }
        for (int i = 0; i < M_COUNT; ++i) { pm[i] = q; q += msz[i]; }
        const int b0 = (int)((long)B * ch / nch), b1 = (int)((long)B * (ch + 1) / nch);
        rnn_backward_range(c, lw, misc, x, t, b0, b1, L, gy, pl, pm, gx, y_out);
    }
    for (int ch = 0; ch < nch; ++ch) {
        const float *q = pool + total * ch;
        for (int i = 0; i < nlw; ++i) { for (size_t k = 0; k < lsz[i]; ++k) glw[i][k] += q[k]; q += lsz[i]; }
        for (int i = 0; i < M_COUNT; ++i) { for (size_t k = 0; k < msz[i]; ++k) gmisc[i][k] += q[k]; q += msz[i]; }
    }
    free(pool);
}

/* ------------------------------------------------------------------------------------------ */
/* Mixed diffusion math                                                                       */
/* ------------------------------------------------------------------------------------------ */
typedef struct {
    int32_t n_num, n_cat, sumK, L, T;
    const int32_t *K;           /* (n_cat) */
    /* (T,) tables, fp32 as registered by mixed_diffusion.py:113-158 */
    const float *sqrt_ac, *sqrt_1m_ac, *sqrt_recip_ac, *sqrt_recipm1_ac, *post_c1, *post_c2, *post_logvar;
    const float *log_alpha, *log_1m_alpha, *log_cum_alpha, *log_1m_cum_alpha;
} OrDiff;

/* cat_log_sample_categorical (mixed_diffusion.py:318-328) for one slice at one (b,l): first-index argmax */
static inline int gumbel_argmax(const float *logit, const float *u, int K, int stride) {
    int best = 0;
    float bv = -INFINITY;
    for (int k = 0; k < K; ++k) {
        float g = -logf(-logf(u[(size_t)k * stride] + 1e-30f) + 1e-30f);
        float v = g + logit[k];
        if (v > bv) { bv = v; best = k; }
    }
    return best;
}

/* Forward noising for a training batch: gauss_q_sample (:240-248), index_to_log_onehot (:44-51),
 * cat_q_sample (:313-316) and the concatenation mixed_loss does at :387.
 * x0_num: (B,n_num,L) already in [-1,1]; x0_idx: (B,n_cat,L); noise: (B,n_num,L); uniforms: (B,sumK,L)
 * -> x_in: (B, n_num+sumK, L), xt_idx: (B,n_cat,L) */
EXPORT void or_q_sample_mixed(const OrDiff *d, int B, const float *x0_num, const uint8_t *x0_idx, const int64_t *t,
                              const float *noise, const float *uniforms, float *x_in, uint8_t *xt_idx) {
    const int L = d->L, C_io = d->n_num + d->sumK;
    for (int b = 0; b < B; ++b) {
        int64_t tb = t[b];
        for (int c = 0; c < d->n_num; ++c)
            for (int l = 0; l < L; ++l) {
                size_t i = ((size_t)b * d->n_num + c) * L + l;
                x_in[((size_t)b * C_io + c) * L + l] = d->sqrt_ac[tb] * x0_num[i] + d->sqrt_1m_ac[tb] * noise[i];
            }
        int off = 0;
        for (int v = 0; v < d->n_cat; ++v) {
            int K = d->K[v];
            float logK = logf((float)K);
            for (int l = 0; l < L; ++l) {
                int x0 = x0_idx[((size_t)b * d->n_cat + v) * L + l];
                float logit[K];
                for (int k = 0; k < K; ++k) {
                    float lx = (k == x0) ? 0.f : LOG_1E30;
                    logit[k] = log_add_exp(lx + d->log_cum_alpha[tb], d->log_1m_cum_alpha[tb] - logK); /* :264-272 */
                }
                int s = gumbel_argmax(logit, uniforms + ((size_t)b * d->sumK + off) * L + l, K, L);
                if (xt_idx) xt_idx[((size_t)b * d->n_cat + v) * L + l] = (uint8_t)s;
                for (int k = 0; k < K; ++k)
                    x_in[((size_t)b * C_io + d->n_num + off + k) * L + l] = (k == s) ? 0.f : LOG_1E30;
            }
            off += K;
        }
    }
}

/* unnormalised log posterior of one slice at one (b,l): cat_q_posterior :283-291 without :293 */
static inline void cat_posterior_unnormed(const OrDiff *d, int64_t t, int K, const float *log_x0, int xt, float *out) {
    float logK = logf((float)K);
    int64_t tm1 = t - 1 < 0 ? 0 : t - 1;
    for (int k = 0; k < K; ++k) {
        float ev = (t == 0) ? log_x0[k]
                            : log_add_exp(log_x0[k] + d->log_cum_alpha[tm1], d->log_1m_cum_alpha[tm1] - logK);
        float lxt = (k == xt) ? 0.f : LOG_1E30;
        float w = log_add_exp(lxt + d->log_alpha[t], d->log_1m_alpha[t] - logK); /* :255-262 */
        out[k] = ev + w;
    }
}

static inline void log_softmax_slice(const float *z, int K, int stride, float *out) {
    float m = -INFINITY, s = 0.f;
    for (int k = 0; k < K; ++k) m = fmaxf(m, z[(size_t)k * stride]);
    for (int k = 0; k < K; ++k) s += expf(z[(size_t)k * stride] - m);
    float lse = m + logf(s);
    for (int k = 0; k < K; ++k) out[k] = z[(size_t)k * stride] - lse;
}

/* log theta for all slices of one (b,l) column.  faithful: normalise like the reference (:293 with
 * sliced_logsumexp), else by a direct per-slice logsumexp. */
static void column_log_theta(const OrDiff *d, int64_t t, const float *log_x0_all /*(sumK)*/, const uint8_t *xt /*(n_cat) stride*/,
                             int xt_stride, int faithful, float *out /*(sumK)*/) {
    int off = 0;
    for (int v = 0; v < d->n_cat; ++v) {
        cat_posterior_unnormed(d, t, d->K[v], log_x0_all + off, xt[(size_t)v * xt_stride], out + off);
        off += d->K[v];
    }
    float lse[d->n_cat];
    if (faithful) {
        int64_t offs[d->n_cat + 1];
        offs[0] = 0;
        for (int v = 0; v < d->n_cat; ++v) offs[v + 1] = offs[v] + d->K[v];
        sliced_lse_faithful_col(out, 1, offs, d->n_cat, lse);
    } else {
        off = 0;
        for (int v = 0; v < d->n_cat; ++v) {
            float m = -INFINITY, s = 0.f;
            for (int k = 0; k < d->K[v]; ++k) m = fmaxf(m, out[off + k]);
            for (int k = 0; k < d->K[v]; ++k) s += expf(out[off + k] - m);
            lse[v] = m + logf(s);
            off += d->K[v];
        }
    }
    off = 0;
    for (int v = 0; v < d->n_cat; ++v) {
        for (int k = 0; k < d->K[v]; ++k) out[off + k] -= lse[v];
        off += d->K[v];
    }
}

/* One reverse update: gauss_p_sample (:412-423 -> :196-238) and cat_p_sample (:307-311 -> :274-305, :318-328).
 * model_out: (B, n_num+sumK, L); z: (B,n_num,L); idx: (B,n_cat,L); t: (B) (the sampler passes one value);
 * noise: (B,n_num,L) (ignored where t==0); uniforms: (B,sumK,L).
 * outputs (any may be NULL): z_new, idx_new, x0 (post-clamp), mean, log_theta (B,sumK,L). */
EXPORT void or_reverse_step_mixed(const OrDiff *d, int B, const float *model_out, const float *z, const uint8_t *idx,
                                  const int64_t *t, const float *noise, const float *uniforms, int faithful, float *z_new,
                                  uint8_t *idx_new, float *x0_out, float *mean_out, float *log_theta_out) {
    const int L = d->L, C_io = d->n_num + d->sumK;
    for (int b = 0; b < B; ++b) {
        int64_t tb = t[b];
        for (int c = 0; c < d->n_num; ++c)
            for (int l = 0; l < L; ++l) {
                size_t i = ((size_t)b * d->n_num + c) * L + l;
                float eps = model_out[((size_t)b * C_io + c) * L + l];
                float x0 = d->sqrt_recip_ac[tb] * z[i] - d->sqrt_recipm1_ac[tb] * eps;
                x0 = fminf(fmaxf(x0, -1.f), 1.f);
                float mean = d->post_c1[tb] * x0 + d->post_c2[tb] * z[i];
                float nz = tb > 0 ? noise[i] : 0.f;
                if (x0_out) x0_out[i] = x0;
                if (mean_out) mean_out[i] = mean;
                if (z_new) z_new[i] = mean + expf(0.5f * d->post_logvar[tb]) * nz;
            }
        for (int l = 0; l < L; ++l) {
            float lx0[d->sumK > 0 ? d->sumK : 1], lth[d->sumK > 0 ? d->sumK : 1];
            int off = 0;
            for (int v = 0; v < d->n_cat; ++v) {
                log_softmax_slice(model_out + ((size_t)b * C_io + d->n_num + off) * L + l, d->K[v], L, lx0 + off);
                off += d->K[v];
            }
            column_log_theta(d, tb, lx0, idx + ((size_t)b * d->n_cat) * L + l, L, faithful, lth);
            off = 0;
            for (int v = 0; v < d->n_cat; ++v) {
                if (log_theta_out)
                    for (int k = 0; k < d->K[v]; ++k) log_theta_out[((size_t)b * d->sumK + off + k) * L + l] = lth[off + k];
                if (idx_new)
                    idx_new[((size_t)b * d->n_cat + v) * L + l] =
                        (uint8_t)gumbel_argmax(lth + off, uniforms + ((size_t)b * d->sumK + off) * L + l, d->K[v], L);
                off += d->K[v];
            }
        }
    }
}

/* Smallest top-2 gap of the Gumbel-perturbed logits per (b, variable, l): the margin the parity
 * gate of BASELINE.json's north_star is stated on. */
EXPORT void or_gumbel_margin(const OrDiff *d, int B, const float *log_theta, const float *uniforms, float *margin) {
    const int L = d->L;
    for (int b = 0; b < B; ++b) {
        int off = 0;
        for (int v = 0; v < d->n_cat; ++v) {
            for (int l = 0; l < L; ++l) {
                float best = -INFINITY, second = -INFINITY;
                for (int k = 0; k < d->K[v]; ++k) {
                    size_t i = ((size_t)b * d->sumK + off + k) * L + l;
                    float val = -logf(-logf(uniforms[i] + 1e-30f) + 1e-30f) + log_theta[i];
                    if (val > best) { second = best; best = val; }
                    else if (val > second) second = val;
                }
                margin[((size_t)b * d->n_cat + v) * L + l] = best - second;
            }
            off += d->K[v];
        }
    }
}

/* mixed_loss after the denoiser (:390-401): cat_loss (:358-362 -> :341-356, :330-339) / n_cat, gauss_loss
 * (:364-368), and the analytic gradient w.r.t. model_out.
 * outputs (NULL allowed): loss (1), grad (B,C_io,L), Lt (B), kl_prior (B), log_true / log_model (B,sumK,L). */
EXPORT void or_mixed_loss(const OrDiff *d, int B, const float *model_out, const float *noise, const uint8_t *x0_idx,
                          const uint8_t *xt_idx, const int64_t *t, float loss_lambda, int faithful, float *loss_out,
                          float *grad, float *Lt_out, float *klp_out, float *log_true_out, float *log_model_out,
                          float *loss_gauss_out) {
    const int L = d->L, C_io = d->n_num + d->sumK, SK = d->sumK > 0 ? d->sumK : 1;
    double gauss_sum = 0.0, multi_sum = 0.0;
    for (int b = 0; b < B; ++b) {
        int64_t tb = t[b];
        double per_b = 0.0;
        for (int c = 0; c < d->n_num; ++c)
            for (int l = 0; l < L; ++l) {
                float diff = model_out[((size_t)b * C_io + c) * L + l] - noise[((size_t)b * d->n_num + c) * L + l];
                per_b += (double)diff * diff;
                if (grad) grad[((size_t)b * C_io + c) * L + l] = 2.0f * diff / ((float)B * d->n_num * L);
            }
        if (d->n_num > 0) gauss_sum += per_b / ((double)d->n_num * L);
        double kl_b = 0.0, nll_b = 0.0, prior_b = 0.0;
        for (int l = 0; l < L; ++l) {
            float lx0[SK], lx0hat[SK], ltrue[SK], lmodel[SK];
            int off = 0;
            for (int v = 0; v < d->n_cat; ++v) {
                int x0 = x0_idx[((size_t)b * d->n_cat + v) * L + l];
                for (int k = 0; k < d->K[v]; ++k) lx0[off + k] = (k == x0) ? 0.f : LOG_1E30;
                log_softmax_slice(model_out + ((size_t)b * C_io + d->n_num + off) * L + l, d->K[v], L, lx0hat + off);
                off += d->K[v];
            }
            const uint8_t *xt = xt_idx + ((size_t)b * d->n_cat) * L + l;
            column_log_theta(d, tb, lx0, xt, L, faithful, ltrue);
            column_log_theta(d, tb, lx0hat, xt, L, faithful, lmodel);
            off = 0;
            for (int v = 0; v < d->n_cat; ++v) {
                int K = d->K[v];
                float logK = logf((float)K);
                for (int k = 0; k < K; ++k) {
                    size_t o = ((size_t)b * d->sumK + off + k) * L + l;
                    if (log_true_out) log_true_out[o] = ltrue[off + k];
                    if (log_model_out) log_model_out[o] = lmodel[off + k];
                    kl_b += (double)(expf(ltrue[off + k]) * (ltrue[off + k] - lmodel[off + k]));   /* :251-253 */
                    nll_b -= (double)(expf(lx0[off + k]) * lmodel[off + k]);                      /* :56-57, :351 */
                    float lq = log_add_exp(lx0[off + k] + d->log_cum_alpha[d->T - 1], d->log_1m_cum_alpha[d->T - 1] - logK);
                    prior_b += (double)(expf(lq) * (lq - (-logK)));                              /* :330-339 */
                }
                if (grad) {
                    /* g_k = s (theta_k sum_j tgt_j - tgt_k) / pi_k ; dz = scale * p (g - sum g p) */
                    float p[K], pi[K], gk[K];
                    float s = (tb == 0) ? 1.0f : expf(d->log_cum_alpha[tb - 1 < 0 ? 0 : tb - 1]);
                    float tgt_sum = 0.f, gp = 0.f;
                    for (int k = 0; k < K; ++k) {
                        p[k] = expf(lx0hat[off + k]);
                        pi[k] = (tb == 0) ? p[k] : s * p[k] + (1.0f - s) / K;
                        tgt_sum += (tb == 0) ? expf(lx0[off + k]) : expf(ltrue[off + k]);
                    }
                    for (int k = 0; k < K; ++k) {
                        float tgt = (tb == 0) ? expf(lx0[off + k]) : expf(ltrue[off + k]);
                        gk[k] = s * (expf(lmodel[off + k]) * tgt_sum - tgt) / pi[k];
                        gp += gk[k] * p[k];
                    }
                    float scale = loss_lambda * (float)d->T / ((float)B * d->n_cat);
                    for (int k = 0; k < K; ++k)
                        grad[((size_t)b * C_io + d->n_num + off + k) * L + l] = scale * p[k] * (gk[k] - gp);
                }
                off += K;
            }
        }
        float Lt = (tb == 0) ? (float)nll_b : (float)kl_b;
        if (Lt_out) Lt_out[b] = Lt;
        if (klp_out) klp_out[b] = (float)prior_b;
        if (d->n_cat > 0) multi_sum += ((double)Lt * d->T + prior_b) / d->n_cat;
    }
    float lg = d->n_num > 0 ? (float)(gauss_sum / B) : 0.f;
    float lm = d->n_cat > 0 ? (float)(multi_sum / B) : 0.f;
    if (loss_gauss_out) *loss_gauss_out = lg;
    if (loss_out) *loss_out = loss_lambda * lm + lg;
}

/* Prior draw of the sampler (mixed_diffusion.py:434-435): Gumbel-max over uniform logits. */
EXPORT void or_prior_categorical(const OrDiff *d, int B, const float *uniforms, uint8_t *idx) {
    const int L = d->L;
    for (int b = 0; b < B; ++b) {
        int off = 0;
        for (int v = 0; v < d->n_cat; ++v) {
            float zero[d->K[v]];
            memset(zero, 0, sizeof zero);
            for (int l = 0; l < L; ++l)
                idx[((size_t)b * d->n_cat + v) * L + l] =
                    (uint8_t)gumbel_argmax(zero, uniforms + ((size_t)b * d->sumK + off) * L + l, d->K[v], L);
            off += d->K[v];
        }
    }
}

/* Denoiser input from sampler state (mixed_diffusion.py:439 with :44-51): cat([z, log-one-hot(idx)]). */
EXPORT void or_build_x_in(const OrDiff *d, int B, const float *z, const uint8_t *idx, float *x_in) {
    const int L = d->L, C_io = d->n_num + d->sumK;
    for (int b = 0; b < B; ++b) {
        memcpy(x_in + (size_t)b * C_io * L, z + (size_t)b * d->n_num * L, sizeof(float) * (size_t)d->n_num * L);
        int off = 0;
        for (int v = 0; v < d->n_cat; ++v) {
            for (int k = 0; k < d->K[v]; ++k)
                for (int l = 0; l < L; ++l)
                    x_in[((size_t)b * C_io + d->n_num + off + k) * L + l] =
                        (idx[((size_t)b * d->n_cat + v) * L + l] == k) ? 0.f : LOG_1E30;
            off += d->K[v];
        }
    }
}

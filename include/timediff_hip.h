/*
 * timediff_hip.h -- C ABI of libtimediff_hip.so: the MI355X (gfx950) implementation of TimeDiff's
 * ETDiff hot path (denoiser forward/backward, mixed-diffusion forward noising, reverse update,
 * loss, T-step sampler).
 *
 * The reference (MuhangTian/TimeDiff) is pure Python and has no FFI of its own; each entry point
 * cites the Python symbol (file:line under the reference tree) whose arithmetic it replaces.  The
 * host-side mirror that keeps the reference's class interfaces lives in the timediff_amd Python package and
 * binds these symbols with ctypes (INTEGRATION.md shows the stub).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter name ends in `_host`;
 *   - tensors are dense row-major in the reference's (torch) layouts; weights are read in torch's
 *     native layout (LSTM gate rows i,f,g,o; GRU gate rows r,z,n);
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream); every call only
 *     enqueues work and takes its scratch through the `workspace` arguments sized by the matching
 *     *_bytes() query.  Exceptions, each once per process: td_sample_chain instantiates and caches a hipGraph
 *     (released by td_sample_chain_release); td_profile_enable creates HIP events; the first td_rnn_pack_weights
 *     allocates 64 bytes of device-writable host memory (td_check_async_errors); td_rnn_backward creates two streams and
 *     19 events per device at its first small-batch call (td_set_side_gemm).  The library links no vendor
 *     math library: every GEMM on the path is a kernel of this tree;
 *   - limits of this build: hidden <= 1024, c_in <= 32, c_out <= 32, at most TD_MAX_CAT discrete variables
 *     of at most TD_MAX_CLASSES classes each (the reference accepts any K_i, mixed_diffusion.py:44-51);
 *   - return value: TD_OK or an error code; td_last_error() returns the message (thread-local).
 */
#ifndef TIMEDIFF_HIP_H
#define TIMEDIFF_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define TD_API __attribute__((visibility("default")))
#else
#define TD_API
#endif

enum { TD_OK = 0, TD_ERR_INVALID = 1, TD_ERR_HIP = 2, TD_ERR_UNSUPPORTED = 3 };

TD_API int td_version(void);
TD_API const char *td_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Denoiser: models/ETDiff/blocks.py RNN (:300-341) = time_emb (:324-329, SinusoidalPosEmb :12-24)
 * + RNNBlock (:264-297: nn.LSTM / nn.GRU, LayerNorm, FiLM) + fc.
 * ------------------------------------------------------------------------------------------ */
typedef struct td_rnn_config {
    int32_t c_in;      /* RNN(input_channels)                       */
    int32_t hidden;    /* RNN(hidden_channels)                      */
    int32_t c_out;     /* RNN(output_channels)                      */
    int32_t layers;    /* RNN(layers)                               */
    int32_t dirs;      /* 2 if bidirectional else 1                 */
    int32_t gru;       /* model == "gru"                            */
    int32_t embed_dim; /* RNN(embed_dim), SinusoidalPosEmb width    */
    int32_t time_dim;  /* RNN(time_dim)                             */
} td_rnn_config;

/* Index of each tensor in the `misc_host` pointer table (state_dict names in comments). */
enum {
    TD_TE_W1 = 0, /* time_emb.1.weight      (time_dim, embed_dim) */
    TD_TE_B1,     /* time_emb.1.bias                                */
    TD_TE_W2,     /* time_emb.3.weight      (time_dim, time_dim)  */
    TD_TE_B2,     /* time_emb.3.bias                                */
    TD_FILM_W,    /* rnn.time_mlp.1.weight  (2*D*H, time_dim)     */
    TD_FILM_B,    /* rnn.time_mlp.1.bias                            */
    TD_LN_W,      /* rnn.norm.weight        (D*H)                 */
    TD_LN_B,      /* rnn.norm.bias                                  */
    TD_FC_W,      /* fc.weight              (c_out, D*H)          */
    TD_FC_B,      /* fc.bias                                        */
    TD_MISC_COUNT
};

/* Bytes of the MFMA-ordered bf16 copy of the recurrent weights (a derived cache; rebuild after
 * every optimizer step) and of the forward workspace for a (B, L) batch. */
TD_API size_t td_rnn_packed_bytes(const td_rnn_config *cfg);
TD_API size_t td_rnn_workspace_bytes(const td_rnn_config *cfg, int32_t batch, int32_t seq_len);

/* layer_weights_host[(layer*dirs + dir)*4 + {0: weight_ih, 1: weight_hh, 2: bias_ih, 3: bias_hh}]
 * (rnn.rnn.weight_ih_l{k}[_reverse] ...): host array of device pointers to fp32 tensors. */
TD_API int td_rnn_pack_weights(const td_rnn_config *cfg, const float *const *layer_weights_host, void *packed, void *stream);

/* blocks.py:338 + :285: film[i, :] = Linear(SiLU(time_emb(t[i]))) for i < n_t; film is (n_t, 2*D*H):
 * scale = film[:, :D*H], shift = film[:, D*H:].  temb_out (n_t, time_dim) may be NULL. */
TD_API int td_rnn_time_film(const td_rnn_config *cfg, const float *const *misc_host, const int64_t *t, int32_t n_t,
                     float *film, float *temb_out, void *stream);
/* Backward of td_rnn_time_film (autograd through blocks.py:324-329, :285 as run by accelerator.backward, et_diff.py:237):
 * given grad_film (n_t, 2*D*H), ACCUMULATES the gradients of the six time-embedding / FiLM parameters into
 * grad_misc[TD_TE_W1 .. TD_FILM_B] (same table layout as misc_host; the other entries are not touched).  The forward
 * values are recomputed from t.  workspace: td_rnn_time_film_backward_workspace_bytes(cfg, n_t) bytes. */
TD_API size_t td_rnn_time_film_backward_workspace_bytes(const td_rnn_config *cfg, int32_t n_t);
TD_API int td_rnn_time_film_backward(const td_rnn_config *cfg, const float *const *misc_host, const int64_t *t, int32_t n_t,
                              const float *grad_film, float *const *grad_misc_host, void *workspace, void *stream);

/* blocks.py:333-341 after the time embedding: y = fc(FiLM(LayerNorm(BRNN(x^T))))^T.
 *   x: (B, c_in, L) fp32    y: (B, c_out, L) fp32
 *   film: rows of 2*D*H; row used for batch element b is
 *         film_row == NULL ? b * (film_rows > 1) : film_row[0]   (device scalar: the sampler's step counter)
 * Recurrent GEMMs run on bf16 MFMA with fp32 accumulate, fp32 cell state and an fp32-equivalent
 * (split-bf16) first-layer input projection. */
TD_API int td_rnn_forward(const td_rnn_config *cfg, const void *packed, const float *const *misc_host, const float *x,
                   const float *film, int32_t film_rows, const int32_t *film_row, int32_t batch, int32_t seq_len,
                   float *y, void *workspace, void *stream);

/* Parity/debug: the last layer's output (what nn.LSTM returns, blocks.py:289-290) as (B, L, D*H)
 * fp32, decoded from the workspace left by td_rnn_forward. */
TD_API int td_rnn_read_hidden(const td_rnn_config *cfg, const void *workspace, int32_t batch, int32_t seq_len, float *hidden,
                       void *stream);

/* Training (accelerator.backward(loss), et_diff.py:237, through blocks.py:333-341).
 * td_rnn_forward_train = td_rnn_forward with one film row per batch element, keeping in `workspace`
 * (td_rnn_train_workspace_bytes) every layer's output, the gate activations and bf16 row-major copies of
 * the activations.  td_rnn_backward consumes that workspace:
 *   grad_y (B, c_out, L) -> grad_layer_host[...] (same indexing as layer_weights_host, fp32 torch layout,
 *   ACCUMULATED into), grad_misc_host[TD_LN_W, TD_LN_B, TD_FC_W, TD_FC_B] (accumulated; other slots
 *   ignored), grad_film (B, 2*D*H) (overwritten; the time-embedding MLP's backward is the caller's).
 * packT: transposed bf16 weight fragments (td_rnn_packT_bytes / td_rnn_pack_weights_T), a derived cache
 * like `packed`.  Gradients are computed from bf16-rounded activations and gate gradients with fp32
 * accumulation; float atomics make the last bits run-dependent. */
TD_API size_t td_rnn_train_workspace_bytes(const td_rnn_config *cfg, int32_t batch, int32_t seq_len);
/* Inter-layer dropout of nn.LSTM / nn.GRU (`dropout=` of blocks.py:268-270; etdiff_train.py --dropout): in training
 * every layer's output but the top one's is multiplied by a Bernoulli(1 - p) mask / (1 - p) on its way to the next layer.
 * Process-wide, like the kernel-selection knobs: td_rnn_train_workspace_bytes, td_rnn_forward_train and
 * td_rnn_backward[_cb] read it (a backward pass must see the (p, seed) of its forward pass: the mask is regenerated
 * from a counter-based generator, not stored); inference entry points never drop.  p is quantised to 1/65536.
 * td_rnn_dropout_mask writes layer `layer`'s mask as (B, L, D*H) fp32 (entries 0 or 1/(1-p)) for parity checks. */
TD_API int td_rnn_set_dropout(float p, uint64_t seed);
TD_API int td_rnn_dropout_mask(const td_rnn_config *cfg, int32_t batch, int32_t seq_len, int32_t layer, float *mask, void *stream);
TD_API size_t td_rnn_packT_bytes(const td_rnn_config *cfg);
TD_API int td_rnn_pack_weights_T(const td_rnn_config *cfg, const float *const *layer_weights_host, void *packT, void *stream);
TD_API int td_rnn_forward_train(const td_rnn_config *cfg, const void *packed, const float *const *misc_host, const float *x,
                         const float *film, int32_t batch, int32_t seq_len, float *y, void *workspace, void *stream);
TD_API int td_rnn_backward(const td_rnn_config *cfg, const void *packT, const float *const *misc_host, const float *film,
                    const float *grad_y, int32_t batch, int32_t seq_len, void *workspace, float *const *grad_layer_host,
                    float *const *grad_misc_host, float *grad_film, void *stream);
/* td_rnn_backward with a host callback for data-parallel training (the reference gets the same overlap from DDP's
 * bucketed all-reduce inside accelerator.backward, et_diff.py:237): `layer_done(l, user)` is called on the calling
 * thread right after the last kernel that writes layer l's parameter gradients has been ENQUEUED on `stream`
 * (l = layers first, for the LayerNorm / fc gradients; then layers-1 .. 0).  The callback typically records an
 * event on `stream` and starts that layer's gradient all-reduce on another stream behind it, while the
 * layers below are still being differentiated.  layer_done == NULL: plain td_rnn_backward. */
typedef void (*td_layer_done_fn)(int32_t layer, void *user);
TD_API int td_rnn_backward_cb(const td_rnn_config *cfg, const void *packT, const float *const *misc_host, const float *film,
                       const float *grad_y, int32_t batch, int32_t seq_len, void *workspace, float *const *grad_layer_host,
                       float *const *grad_misc_host, float *grad_film, td_layer_done_fn layer_done, void *user, void *stream);

/* td_rnn_backward_cb followed by td_rnn_time_film_backward(cfg, misc, t, batch, grad_film, grad_misc, film_workspace) in ONE
 * call: the whole backward pass of RNN.forward (blocks.py:333-341 under accelerator.backward, et_diff.py:237) for one film row
 * per batch element.  grad_misc_host holds all ten entries (the six time-embedding / FiLM-linear parameters are accumulated
 * as td_rnn_time_film_backward does); grad_film is scratch of the caller's ((batch, 2*D*H), complete when the call's work
 * is).  What the single call buys: the time-embedding backward runs beside the BPTT sweeps (td_set_side_gemm) instead of
 * behind them.  film_workspace: td_rnn_time_film_backward_workspace_bytes(cfg, batch) bytes. */
TD_API int td_rnn_backward_all(const td_rnn_config *cfg, const void *packT, const float *const *misc_host, const float *film,
                        const int64_t *t, const float *grad_y, int32_t batch, int32_t seq_len, void *workspace,
                        void *film_workspace, float *const *grad_layer_host, float *const *grad_misc_host, float *grad_film,
                        td_layer_done_fn layer_done, void *user, void *stream);

/* ------------------------------------------------------------------------------------------
 * Mixed diffusion: models/ETDiff/mixed_diffusion.py MixedDiffusion (:68-454); n_cat = 0 gives
 * models/ETDiff/gaussian_diffusion.py GaussianDiffusion.
 * ------------------------------------------------------------------------------------------ */
enum { /* rows of the (TD_TAB_COUNT, T) fp32 `tables` array = buffers registered at mixed_diffusion.py:113-158 */
    TD_TAB_SQRT_AC = 0,      /* sqrt_alphas_cumprod            */
    TD_TAB_SQRT_1M_AC,       /* sqrt_one_minus_alphas_cumprod  */
    TD_TAB_SQRT_RECIP_AC,    /* sqrt_recip_alphas_cumprod      */
    TD_TAB_SQRT_RECIPM1_AC,  /* sqrt_recipm1_alphas_cumprod    */
    TD_TAB_POST_C1,          /* posterior_mean_coef1           */
    TD_TAB_POST_C2,          /* posterior_mean_coef2           */
    TD_TAB_POST_LOGVAR,      /* posterior_log_variance_clipped */
    TD_TAB_LOG_ALPHA,        /* log_alpha                      */
    TD_TAB_LOG_1M_ALPHA,     /* log_1_min_alpha                */
    TD_TAB_LOG_CUM_ALPHA,    /* log_cumprod_alpha              */
    TD_TAB_LOG_1M_CUM_ALPHA, /* log_1_min_cumprod_alpha        */
    TD_TAB_COUNT
};

#define TD_MAX_CAT 32    /* discrete variables per model */
#define TD_MAX_CLASSES 32 /* classes per discrete variable (more than 8: scalar kernels instead of 4 positions per thread) */

typedef struct td_diffusion_config {
    int32_t n_num;               /* continuous channels                                   */
    int32_t n_cat;               /* discrete variables                                    */
    int32_t seq_len;             /* L                                                     */
    int32_t timesteps;           /* T                                                     */
    int32_t num_classes[TD_MAX_CAT]; /* K_i (categorical_num_classes)                     */
    int32_t num_index[TD_MAX_CAT * 2]; /* numerical_features_indices (dataset channel)    */
    int32_t cat_index[TD_MAX_CAT];   /* categorical_features_indices (dataset channel)    */
    int32_t data_channels;       /* channels of a dataset tensor / of sample()'s result   */
    int32_t raw_range;           /* 0: auto_normalize=True, the continuous channels are mapped [0, 1] -> [-1, 1] on the way
                                    in and back on the way out (mixed_diffusion.py:160-162); 1: auto_normalize=False, identity.
                                    (Occupies what was padding in front of the pointer: zero-initialised configs keep meaning 0.) */
    const float *tables;         /* device, (TD_TAB_COUNT, T)                             */
} td_diffusion_config;

/* Noise sources.  A NULL `noise` / `uniforms` pointer selects the in-kernel Philox4x32-10 stream
 * (seed, patient = patient0 + b, position l, tag); the layout of the counters is in
 * timediff_amd/csrc/common.h and oracle/etdiff_oracle.c.  Parity runs pass the reference's draws:
 *   noise (B, n_num, L), uniforms (B, sum K, L)   [draw order: SURVEY.md section 8c]. */
typedef struct td_rng {
    uint64_t seed;
    uint32_t patient0;
    uint32_t tag;
    const uint64_t *seed_dev;   /* NULL, or a device word ADDED to seed when the kernel runs: a launch captured in a hipGraph
                                   draws fresh noise on every replay (the caller rewrites the word between replays) */
} td_rng;

/* mixed_diffusion.py:403-409 + :375-387: normalize_numericals, gauss_q_sample (:240-248),
 * index_to_log_onehot (:44-51), cat_q_sample (:313-316), concatenation.
 *   data: (B, data_channels, L) fp32, numericals in [0,1], discrete channels hold class ids
 *   t: (B) int64.   Outputs: x_in (B, n_num + sum K, L), x0_idx / xt_idx (B, n_cat, L) uint8,
 *   noise_out (B, n_num, L): the epsilon actually used (needed by the loss). */
TD_API int td_q_sample_mixed(const td_diffusion_config *cfg, int32_t batch, const float *data, const int64_t *t,
                      const float *noise, const float *uniforms, td_rng rng, float *x_in, uint8_t *x0_idx,
                      uint8_t *xt_idx, float *noise_out, void *stream);

/* mixed_diffusion.py:390-401: cat_loss (:358-362, :341-356, :330-339) / n_cat and gauss_loss (:364-368)
 * plus the analytic gradient w.r.t. model_out.  loss: device scalar; grad: (B, n_num + sum K, L) or NULL.
 * `scale` multiplies the gradient (1/gradient_accumulate_every, et_diff.py:234).
 * workspace: td_mixed_loss_workspace_bytes(batch). */
TD_API size_t td_mixed_loss_workspace_bytes(const td_diffusion_config *cfg, int32_t batch);
TD_API int td_mixed_loss(const td_diffusion_config *cfg, int32_t batch, const float *model_out, const float *noise,
                  const uint8_t *x0_idx, const uint8_t *xt_idx, const int64_t *t, float loss_lambda, float scale,
                  float *loss, float *grad, void *workspace, void *stream);

/* td_mixed_loss without the workspace argument.  No counterpart in the reference: it is for callers that cannot
 * allocate device memory (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * The block partials live in an 8 KiB file-scope device array of the library (one copy per GPU, held until the
 * process ends); the call allocates and frees nothing.  Limits: batch * seq_len <= TD_NOALLOC_MAX_POSITIONS
 * (TD_ERR_UNSUPPORTED beyond); calls that overlap on one GPU share the array and race: keep them on one stream.
 * The first call on a device looks the array's address up (hipGetSymbolAddress): make it outside graph capture. */
#define TD_NOALLOC_MAX_POSITIONS (1 << 27)
TD_API int td_mixed_loss_noalloc(const td_diffusion_config *cfg, int32_t batch, const float *model_out, const float *noise,
                          const uint8_t *x0_idx, const uint8_t *xt_idx, const int64_t *t, float loss_lambda, float scale,
                          float *loss, float *grad, void *stream);

/* mixed_diffusion.py:429-435: z_T ~ N(0, I) (B, n_num, L) and the uniform categorical prior as class ids. */
TD_API int td_sample_prior(const td_diffusion_config *cfg, int32_t batch, const float *noise, const float *uniforms,
                    td_rng rng, float *z, uint8_t *idx, void *stream);

/* mixed_diffusion.py:439 with :44-51: denoiser input cat([z, log_onehot(idx)]) as (B, n_num + sum K, L). */
TD_API int td_build_denoiser_input(const td_diffusion_config *cfg, int32_t batch, const float *z, const uint8_t *idx,
                            float *x_in, void *stream);

/* One reverse update in place: gauss_p_sample (mixed_diffusion.py:412-423 -> :196-238) and
 * cat_p_sample (:307-311 -> :274-305, :318-328).  The step index is `t_host` when t_dev == NULL,
 * else the device scalar t_dev[0] (hipGraph replay).  rng.tag is ignored: the tag is the step index. */
TD_API int td_reverse_step_mixed(const td_diffusion_config *cfg, int32_t batch, const float *model_out, float *z,
                          uint8_t *idx, int32_t t_host, const int32_t *t_dev, const float *noise,
                          const float *uniforms, td_rng rng, void *stream);

/* Parity aid: the posterior terms of one reverse step as the fused kernel computes them, nothing sampled or
 * updated: x0 = clamp(predict_start_from_noise) (mixed_diffusion.py:196-201, :234-235) and mean = posterior mean
 * (:210-218), both (B, n_num, L); log_theta = cat_q_posterior's normalised log-probabilities (:283-295), (B, sum K, L).
 * Same kernel as td_reverse_step_mixed with the sampling replaced by these stores. */
TD_API int td_reverse_step_terms(const td_diffusion_config *cfg, int32_t batch, const float *model_out, const float *z,
                          const uint8_t *idx, int32_t t, float *x0, float *mean, float *log_theta, void *stream);

/* mixed_diffusion.py:448-453 (+ :59-66, :173-179, utils.py:95-99): (z+1)/2 and class ids scattered to
 * dataset channel order, (B, data_channels, L) fp32. */
TD_API int td_decode_samples(const td_diffusion_config *cfg, int32_t batch, const float *z, const uint8_t *idx,
                      float *samples, void *stream);

/* The whole reverse chain, mixed_diffusion.py:425-454 (and gaussian_diffusion.py:133-152 when n_cat = 0):
 * prior draw, T x (denoiser forward + reverse update) replayed as a captured hipGraph whose kernels
 * read the step index from a device counter, decode.  No host round trip inside the chain.
 *   film_table: (T, 2*D*H) from td_rnn_time_film(t = 0..T-1)
 *   noise_host_or_null / uniforms: optional externally supplied draws, (T+1, ...) in loop order with
 *   index 0 = prior, index 1 + s = step t = T-1-s; NULL = Philox.
 *   samples: (B, data_channels, L) fp32 on the device. */
TD_API size_t td_sample_chain_workspace_bytes(const td_rnn_config *rnn, const td_diffusion_config *cfg, int32_t batch);
TD_API int td_sample_chain(const td_rnn_config *rnn, const td_diffusion_config *cfg, const void *packed,
                    const float *const *misc_host, const float *film_table, int32_t batch, const float *noise,
                    const float *uniforms, td_rng rng, int32_t use_graph, float *samples, void *workspace,
                    void *stream);

/* Releases the cached step graph of td_sample_chain (waits for the stream it was launched on).
 * No counterpart in the reference. */
TD_API int td_sample_chain_release(void);

/* What changes from one optimizer step to the next, as a record in DEVICE memory: td_clip_adam_ema_rec reads it when the
 * kernel runs, so a whole training step captured in a hipGraph is replayed with fresh values (the caller
 * rewrites it with td_step_record_set before each replay; `seed` is for td_rng.seed_dev).  No counterpart
 * in the reference. */
typedef struct td_step_record {
    uint64_t seed;
    float step_size;      /* lr / (1 - beta1^step) */
    float inv_sqrt_bc2;   /* 1 / sqrt(1 - beta2^step) */
    float ema_decay;
    int32_t ema_mode;
    uint64_t reserved;
} td_step_record;

/* ------------------------------------------------------------------------------------------
 * Optimizer step of ETDiff.train (models/ETDiff/et_diff.py:239 clip_grad_norm_(1.0), :75/:247
 * torch.optim.Adam(lr, betas), :254 EMA update) over FLAT fp32 buffers of all parameters.
 *   td_grad_norm: norm[0] = || grad ||_2 (device scalar).
 *   td_clip_adam_ema: g *= min(1, max_norm / (norm + 1e-6)) (skipped when max_norm <= 0), Adam with bias
 *   correction for 1-based `step`, then ema_mode 0: EMA untouched, 1: ema = w, 2: ema = decay*ema + (1-decay)*w.
 * ------------------------------------------------------------------------------------------ */
TD_API size_t td_grad_norm_workspace_bytes(void);
TD_API int td_grad_norm(const float *grad, size_t n, float *norm, void *workspace, void *stream);
TD_API int td_clip_adam_ema(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *ema, size_t n,
                     const float *norm, float max_norm, float lr, float beta1, float beta2, float eps, int32_t step,
                     int32_t ema_mode, float ema_decay, void *stream);
/* The same pass with step size, bias correction and EMA mode / decay read from a td_step_record in device memory when the
 * kernel runs (hipGraph replay); td_step_record_set computes the record exactly as td_clip_adam_ema computes its own
 * values from (lr, betas, step) and writes it to device memory by a one-thread kernel (stream-ordered, no host buffer). */
TD_API int td_step_record_set(td_step_record *record_dev, uint64_t seed, float lr, float beta1, float beta2, int32_t step,
                       int32_t ema_mode, float ema_decay, void *stream);
TD_API int td_clip_adam_ema_rec(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, float *ema, size_t n,
                         const float *norm, float max_norm, float beta1, float beta2, float eps,
                         const td_step_record *record, void *stream);

/* ---- TSTR classifier: the nearest downstream consumer of the same recurrent kernels --------------------------------
 * evaluate/nn_classifier.py:89-127 (class RNN: bidirectional nn.LSTM / nn.GRU over (B, C, L), final hidden state of each
 * direction -> Linear(2H, 1) -> Sigmoid) and :140-165 (training with F.binary_cross_entropy).  cfg: c_in = channels,
 * hidden, layers, dirs, gru as for the denoiser (c_out, embed_dim, time_dim are not used: any valid values); weights are
 * packed with td_rnn_pack_weights / td_rnn_pack_weights_T.
 *   td_cls_forward: x (B, c_in, L) -> prob (B); keeps in `workspace` (td_cls_workspace_bytes) what the backward needs.
 *   td_cls_backward: gradient w.r.t. every parameter of mean BCE(prob, y) (y given), or of a loss whose gradient w.r.t.
 *     prob is grad_prob (y NULL), ACCUMULATED into grad_layer (4 pointers per (layer, direction): w_ih, w_hh, b_ih, b_hh in
 *     torch layout), grad_w (D*H) and grad_b (1). */
TD_API size_t td_cls_workspace_bytes(const td_rnn_config *cfg, int32_t batch, int32_t seq_len);
TD_API int td_cls_forward(const td_rnn_config *cfg, const void *packed, const float *x, const float *w, const float *b,
                   int32_t batch, int32_t seq_len, float *prob, void *workspace, void *stream);
TD_API int td_cls_backward(const td_rnn_config *cfg, const void *packT, const float *w, const float *prob, const float *y,
                    const float *grad_prob, int32_t batch, int32_t seq_len, void *workspace, float *const *grad_layer,
                    float *grad_w, float *grad_b, void *stream);

/* Kernel-selection knobs for tests and measurements, no counterpart in the reference (process-wide, not
 * thread-safe against concurrent launches).
 *   td_set_layer_kernel: which recurrent-layer kernel family runs hidden sizes <= 256.  AUTO picks by batch
 *     (csrc/rnn.hip: the cluster-persistent kernel of rnn_cluster.hip up to 2048 bidirectional sequences at hidden
 *     193..256 -- weights resident in registers, 8 workgroups per 32 sequences exchanging h_t through L2 --, the
 *     unit-split kernel of rnn_wide.hip below ~8192 sequences, the batch-block kernel of rnn_layer.h above); BATCH /
 *     UNITS / CLUSTER force one family (UNITS needs hidden > 32; CLUSTER is an error where it cannot run).  The same
 *     mode steers the BPTT sweep of td_rnn_backward: AUTO / CLUSTER take the cluster sweep of rnn_cluster_bwd.hip
 *     where it can run (hidden 193..256, at most 16 batch blocks per cluster = 8192 bidirectional sequences), BATCH /
 *     UNITS the single-workgroup sweep. */
enum { TD_LAYER_KERNEL_AUTO = 0, TD_LAYER_KERNEL_BATCH = 1, TD_LAYER_KERNEL_UNITS = 2, TD_LAYER_KERNEL_CLUSTER = 3 };
TD_API int td_set_layer_kernel(int32_t mode);
/* Which family a (cfg, batch, seq_len) launch takes under the current mode: TD_LAYER_KERNEL_BATCH, _UNITS or _CLUSTER
 * (-1: bad cfg, or a forced family that cannot run this shape). */
TD_API int td_rnn_layer_kernel_choice(const td_rnn_config *cfg, int32_t batch, int32_t seq_len);
/* The sampler's head (LayerNorm + FiLM + fc with one FiLM row for the whole batch) fused into the top layer's kernel: the
 * layer leaves per-direction partial projections and sum h, sum h^2 instead of its activations (csrc/rnn.hip).  AUTO:
 * td_sample_chain fuses wherever the batch-block kernel family runs the layers (large batches, at least two layers),
 * td_rnn_forward never does; OFF: nobody does; ON: td_rnn_forward does too where it can (one FiLM row: film_rows == 1 or
 * film_row given) -- td_rnn_read_hidden then has no activations of the last layer to read. */
enum { TD_HEAD_FUSION_AUTO = 0, TD_HEAD_FUSION_OFF = 1, TD_HEAD_FUSION_ON = 2 };
TD_API int td_set_head_fusion(int32_t mode);
/* td_rnn_backward without a layer_done callback runs a layer's weight-gradient GEMMs on a stream of the library's own (and
 * the head's parameter gradients with the time-embedding backward on a second one), beside
 * the BPTT sweeps of the layers below (forked behind an event on the caller's stream after that layer's sweep, joined before
 * the call returns its last enqueue: the caller's stream order is all a caller needs; a graph capture of the caller's
 * stream captures the fork and join), where the sweep leaves most of the chip idle (small batches).  on = 0: everything on
 * the caller's stream (A/B measurements, tests).  The two streams and their events are created at the first such call on a device.
 * No counterpart in the reference. */
TD_API int td_set_side_gemm(int32_t on);
/* The unit-split family's layers (forward and BPTT sweep) as one launch per time step instead of one persistent launch per
 * layer.  AUTO: hidden sizes above 256 at batches of at most 128 (batch block, direction) pairs, where the persistent
 * kernels leave most of the chip idle behind one CU's L2 port (77 -> 12 us per step at H = 512, batch 512); ON: every
 * width of whole groups of four unit blocks (H padded to a multiple of 128) whenever that family runs; OFF: never.
 * Bit-identical results either way.  No counterpart in the reference. */
enum { TD_STEP_LAUNCH_AUTO = 0, TD_STEP_LAUNCH_OFF = 1, TD_STEP_LAUNCH_ON = 2 };
TD_API int td_set_step_launch(int32_t mode);
/* 1 where a launch of `batch` sequences would run that way under the current knobs, 0 where not, -1 on a bad configuration. */
TD_API int td_rnn_step_launch_choice(const td_rnn_config *cfg, int32_t batch, int32_t seq_len);

/* No counterpart in the reference.  The cluster-persistent kernels (small batches at hidden 193..256; the one-workgroup form
 * at hidden <= 96 has no hand-off between workgroups and cannot raise this) need every workgroup of a launch resident at once.  Where the device cannot give that (shared with another process, CU mask) a bounded poll
 * inside the kernel times out: the outputs are poisoned with NaN and a word in device-writable host memory is raised.
 * This call reads and clears that word without touching the GPU: TD_OK, or TD_ERR_HIP with a message.  Call it wherever
 * the results of enqueued work have been waited for (the Python layer does, after it reads a loss or copies samples). */
TD_API int td_check_async_errors(void);

/* Measurement aid, no counterpart in the reference: while enabled, every launch of the selected kinds (a mask of
 * TD_PROFILE_*: the recurrent-layer kernels of the layers above the first in their inference form, in their saving form,
 * and the BPTT sweeps of those layers) made outside graph capture is bracketed by HIP events on its own stream;
 * td_profile_read waits for them and returns how many launches were timed and the sum of their durations (ms), then
 * resets.  on = 0 switches it off. */
enum { TD_PROFILE_LAYER_INFER = 1, TD_PROFILE_LAYER_SAVE = 2, TD_PROFILE_BPTT_SWEEP = 4 };
TD_API int td_profile_enable(int32_t on);
TD_API int td_profile_read(int32_t *count, float *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* TIMEDIFF_HIP_H */

// The T-step reverse chain (mixed_diffusion.py:425-454, gaussian_diffusion.py:133-152) as a replayed
// hipGraph: one captured step = {denoiser input + layer-0 packing, recurrent layers, head, reverse
// update}; every kernel reads the step index from a device word, so the same executable graph (8 steps) is replayed
// T / 8 times with no host round trip in between.  Two words, used alternately: step j's kernels read word j & 1, and
// its reverse-update kernel leaves t - 1 in the other word for step j + 1 (a separate decrement launch per step was a
// kernel boundary of its own: 3 % of a reverse step at the MIMIC-III shape).
#include "common.h"

#include <string.h>

int td_rnn_denoise(const td_rnn_config *cfg, const void *packed, const float *const *misc, const float *x,
                   const td_diffusion_config *dcfg, const float *z, const uint8_t *idx, const float *film, int film_rows,
                   const int32_t *film_row, int batch, int L, float *y, void *workspace, hipStream_t stream, bool chain);
int td_rnn_state_frags(const td_rnn_config *cfg, const td_diffusion_config *dcfg, const float *z, const uint8_t *idx, int batch,
                       int L, void *workspace, hipStream_t stream, int *kx0);
int td_launch_reverse_step(const td_diffusion_config *cfg, int batch, const float *mo, float *z, uint8_t *idx, int t_host,
                           const int32_t *t_dev, int32_t *t_next, const float *noise, const float *uniforms, size_t noise_stride,
                           size_t uni_stride, td_rng rng, hipStream_t stream, char *frags, int kx0);
bool td_reverse_fusable(const td_diffusion_config *cfg, const float *mo, const float *z, const uint8_t *idx, const float *noise,
                        const float *uniforms, size_t noise_stride, size_t uni_stride, int kx0);

__global__ void td_set_counter_kernel(int32_t *p, int32_t v) { p[0] = v; }

#define TD_STEPS_PER_GRAPH 8

struct TdChainKey {
    td_rnn_config rnn;
    td_diffusion_config cfg;
    const void *packed, *film, *noise, *uniforms, *samples, *workspace, *misc[TD_MISC_COUNT];
    hipStream_t stream;
    td_rng rng;
    int batch;
};
static struct {
    bool valid;
    TdChainKey key;
    hipGraph_t graph;
    hipGraphExec_t exec;
} g_chain;

extern "C" int td_sample_chain_release(void) {
    if (g_chain.valid) {
        (void)hipStreamSynchronize(g_chain.key.stream);
        (void)hipGraphExecDestroy(g_chain.exec);
        (void)hipGraphDestroy(g_chain.graph);
        g_chain.valid = false;
    }
    return TD_OK;
}

struct TdChainLayout {
    size_t z, idx, x_in, mo, counter, rnn, total;
};

static inline size_t td_align(size_t x) { return (x + 255) & ~(size_t)255; }

static TdChainLayout td_chain_layout(const td_rnn_config *rnn, const td_diffusion_config *cfg, int batch) {
    int sumk = 0;
    for (int v = 0; v < cfg->n_cat; ++v) sumk += cfg->num_classes[v];
    const size_t BL = (size_t)batch * cfg->seq_len;
    TdChainLayout l;
    size_t off = 0;
    l.z = off; off += td_align(BL * (cfg->n_num ? cfg->n_num : 1) * sizeof(float));
    l.idx = off; off += td_align(BL * (cfg->n_cat ? cfg->n_cat : 1));
    l.x_in = off; off += td_align(BL * (cfg->n_num + sumk) * sizeof(float));
    l.mo = off; off += td_align(BL * (cfg->n_num + sumk) * sizeof(float));
    l.counter = off; off += 256;
    l.rnn = off; off += td_align(td_rnn_workspace_bytes(rnn, batch, cfg->seq_len));
    l.total = off;
    return l;
}

extern "C" size_t td_sample_chain_workspace_bytes(const td_rnn_config *rnn, const td_diffusion_config *cfg, int32_t batch) {
    if (!rnn || !cfg || batch <= 0 || cfg->seq_len <= 0) return 0;
    if (td_rnn_workspace_bytes(rnn, batch, cfg->seq_len) == 0) return 0;
    return td_chain_layout(rnn, cfg, batch).total;
}

extern "C" int td_sample_chain(const td_rnn_config *rnn, const td_diffusion_config *cfg, const void *packed,
                               const float *const *misc, const float *film_table, int32_t batch, const float *noise,
                               const float *uniforms, td_rng rng, int32_t use_graph, float *samples, void *workspace,
                               void *stream_) {
    TD_REQUIRE(rnn && cfg && packed && misc && film_table && samples && workspace, "NULL argument");
    TD_REQUIRE(batch > 0, "batch must be positive");
    int sumk = 0;
    for (int v = 0; v < cfg->n_cat && v < TD_MAX_CAT; ++v) sumk += cfg->num_classes[v];
    TD_REQUIRE(rnn->c_in == cfg->n_num + sumk && rnn->c_out == rnn->c_in,
               "denoiser width (%d in, %d out) does not match n_num + sum K = %d", rnn->c_in, rnn->c_out, cfg->n_num + sumk);
    hipStream_t stream = (hipStream_t)stream_;
    TD_REQUIRE(!use_graph || stream != nullptr, "graph capture needs a non-default stream");
    const int L = cfg->seq_len, T = cfg->timesteps;
    const TdChainLayout lay = td_chain_layout(rnn, cfg, batch);
    char *ws = (char *)workspace;
    float *z = (float *)(ws + lay.z);
    uint8_t *idx = (uint8_t *)(ws + lay.idx);
    float *x_in = (float *)(ws + lay.x_in), *mo = (float *)(ws + lay.mo);
    int32_t *counter = (int32_t *)(ws + lay.counter);
    void *rnn_ws = ws + lay.rnn;
    const size_t nstride = (size_t)batch * cfg->n_num * L, ustride = (size_t)batch * sumk * L;

    int rc = td_sample_prior(cfg, batch, noise, uniforms, rng, z, idx, stream);
    if (rc != TD_OK) return rc;
    hipLaunchKernelGGL(td_set_counter_kernel, dim3(1), dim3(1), 0, stream, counter, T - 1);
    TD_LAUNCH_CHECK();

    // The denoiser input (mixed_diffusion.py:439) reaches layer 0 as packed fragments at the start of the denoiser
    // workspace.  Packed once here from the prior; every reverse update then leaves the next step's fragments there
    // itself (one kernel fewer per step: 5 % of a step at the MIMIC-III shape).  States the vectorised update does not
    // take (L % 4, more than 8 classes) are packed by a kernel of their own at the head of every step instead.
    int kx0 = 0;
    rc = td_rnn_state_frags(rnn, cfg, nullptr, nullptr, batch, L, rnn_ws, stream, &kx0);
    if (rc != TD_OK) return rc;
    const bool fused = td_reverse_fusable(cfg, mo, z, idx, noise, uniforms, noise ? nstride : 0, uniforms ? ustride : 0, kx0);
    if (fused) {
        rc = td_rnn_state_frags(rnn, cfg, z, idx, batch, L, rnn_ws, stream, nullptr);
        if (rc != TD_OK) return rc;
    }
    auto enqueue_step = [&](int j) -> int {
        int r;
        int32_t *t_now = counter + (j & 1), *t_next = counter + ((j + 1) & 1);
        // the layers and the head (fused into the top layer's kernel where the batch-block family runs it)
        r = td_rnn_denoise(rnn, packed, misc, nullptr, cfg, fused ? nullptr : z, fused ? nullptr : idx, film_table, T, t_now, batch, L,
                           mo, rnn_ws, stream, true);
        if (r != TD_OK) return r;
        return td_launch_reverse_step(cfg, batch, mo, z, idx, 0, t_now, t_next, noise, uniforms, noise ? nstride : 0,
                                      uniforms ? ustride : 0, rng, stream, fused ? (char *)rnn_ws : nullptr, kx0);
    };

    // One graph launch carries several steps: a launch costs the GPU an idle gap of the order of a millisecond
    // (measured: 36.0 ms per step with one step per launch against 35.0 ms of kernel time).
    // A captured launch of a cluster kernel (rnn_cluster.hip) carries its hand-off tags' launch epoch baked in, and stale
    // tags are told apart only because the launch before it on the same exchange buffer had another epoch: a graph
    // therefore holds at least two denoiser passes (one layer = one cluster launch per pass), and a one-step chain is
    // not captured at all.
    const int steps_per_graph = T >= TD_STEPS_PER_GRAPH ? TD_STEPS_PER_GRAPH : 2;
    if (use_graph && T >= 2) {
        // The instantiated step graph is kept for the next call with identical arguments; a stale one
        // is released only after the stream it ran on has drained.
        TdChainKey key;
        memset(&key, 0, sizeof key);
        key.rnn = *rnn; key.cfg = *cfg; key.packed = packed; key.film = film_table; key.batch = batch;
        key.noise = noise; key.uniforms = uniforms; key.rng = rng; key.samples = samples; key.workspace = workspace;
        key.stream = stream;
        for (int i = 0; i < TD_MISC_COUNT; ++i) key.misc[i] = misc[i];
        if (!(g_chain.valid && memcmp(&g_chain.key, &key, sizeof key) == 0)) {
            td_sample_chain_release();
            hipGraph_t graph = nullptr;
            TD_HIP(hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal));
            for (int k = 0; k < steps_per_graph && rc == TD_OK; ++k) rc = enqueue_step(k);     // an even number of steps: a replay starts at word 0 again
            hipError_t e = hipStreamEndCapture(stream, &graph);
            if (rc != TD_OK) {
                if (graph) (void)hipGraphDestroy(graph);
                return rc;
            }
            TD_HIP(e);
            e = hipGraphInstantiate(&g_chain.exec, graph, nullptr, nullptr, 0);
            if (e != hipSuccess) {
                (void)hipGraphDestroy(graph);
                TD_HIP(e);
            }
            g_chain.graph = graph;
            g_chain.key = key;
            g_chain.valid = true;
        }
        // every step reads its index from the device counter, so T steps = T / steps_per_graph replays of the
        // captured block + the remainder enqueued directly
        for (int s = 0; s < T / steps_per_graph; ++s) TD_HIP(hipGraphLaunch(g_chain.exec, stream));
        for (int s = 0; s < T % steps_per_graph; ++s) {
            rc = enqueue_step(s);
            if (rc != TD_OK) return rc;
        }
    } else {
        for (int s = 0; s < T; ++s) {
            rc = enqueue_step(s);
            if (rc != TD_OK) return rc;
        }
    }
    return td_decode_samples(cfg, batch, z, idx, samples, stream);
}

// Derived geometry of a denoiser configuration (shared by rnn.hip and rnn_train.hip).
#pragma once
#include "common.h"
#include "rnn_layout.h"

// ---------------------------------------------------------------------------------------------
// geometry shared by all entry points
// ---------------------------------------------------------------------------------------------
struct TdRnnGeom {
    int nu, hpad, kh, kx0, dirs, layers;
    TdLayerGeom g0, g1;  // layer 0 / layers above
    size_t layer_off[16];
    // cluster-persistent kernel (rnn_cluster.hip; hidden sizes with nu == TD_CLUSTER_NU): its own register-ordered copy
    // of the weights, appended to the packed cache.  cl_dir[0 / 1]: bytes per direction of layer 0 / of a layer above.
    size_t cl_off[16], cl_dir[2];
    size_t packed_bytes;
};
#define TD_CLUSTER_NU 8

// The unit-split family (rnn_wide.hip, and the wide BPTT sweep of rnn_train.hip) as ONE LAUNCH PER TIME STEP: wide layers
// (more than 8 unit blocks) at batches whose persistent kernels would leave most CUs idle behind one CU's L2 port.
// td_set_step_launch(ON) takes every width of whole groups of four unit blocks (tests, measurements).
int td_step_launch_mode(void);   // api.cpp
#ifndef TD_STEP_LAUNCH_MAX_WG
#define TD_STEP_LAUNCH_MAX_WG 128    // persistent workgroups (batch blocks x directions) up to which AUTO launches per step
                                     // (measured at H = 512 / 1024: 2.8x / 4.9x faster at 16 workgroups, 1.2x at 128)
#endif
static inline bool td_use_step_launch(int nu, int nbb, int dirs) {
    if (nu % 4 != 0) return false;               // the kernels' pipeline works chunks of four k-steps
    const int m = td_step_launch_mode();
    if (m == TD_STEP_LAUNCH_OFF) return false;
    if (m == TD_STEP_LAUNCH_ON) return true;
    return nu > 8 && nbb * dirs <= TD_STEP_LAUNCH_MAX_WG;
}

static inline int td_rnn_geom(const td_rnn_config *cfg, TdRnnGeom *g) {
    TD_REQUIRE(cfg != nullptr, "td_rnn_config is NULL");
    TD_REQUIRE(cfg->dirs == 1 || cfg->dirs == 2, "dirs must be 1 or 2, got %d", cfg->dirs);
    TD_REQUIRE(cfg->layers >= 1 && cfg->layers <= 16, "layers must be in [1,16], got %d", cfg->layers);
    TD_REQUIRE(cfg->hidden >= 1 && cfg->c_in >= 1 && cfg->c_out >= 1, "non-positive channel count");
    TD_REQUIRE(cfg->c_out <= 32, "c_out = %d > 32 is not supported", cfg->c_out);
    TD_REQUIRE(cfg->embed_dim >= 4 && cfg->embed_dim % 2 == 0 && cfg->embed_dim <= 1024, "bad embed_dim %d", cfg->embed_dim);
    TD_REQUIRE(cfg->time_dim >= 1 && cfg->time_dim <= 1024, "bad time_dim %d", cfg->time_dim);
    int nu = td_round_nu(cfg->hidden);
    if (nu < 0) {
        td_set_error("hidden = %d > 1024 is not supported by this build", cfg->hidden);
        return TD_ERR_UNSUPPORTED;
    }
    int kx0 = (cfg->c_in + 15) / 16;
    if (kx0 > 2) {
        td_set_error("c_in = %d > 32 is not supported by this build", cfg->c_in);
        return TD_ERR_UNSUPPORTED;
    }
    g->nu = nu; g->hpad = 32 * nu; g->kh = 2 * nu; g->kx0 = kx0; g->dirs = cfg->dirs; g->layers = cfg->layers;
    g->g0 = td_layer_geom(nu, cfg->gru, 1, kx0);
    g->g1 = td_layer_geom(nu, cfg->gru, 0, cfg->dirs);
    size_t off = 0;
    for (int l = 0; l < cfg->layers; ++l) {
        g->layer_off[l] = off;
        off += (l == 0 ? g->g0.dir_bytes : g->g1.dir_bytes) * cfg->dirs;
    }
    g->cl_dir[0] = g->cl_dir[1] = 0;
    if (nu == TD_CLUSTER_NU || nu <= 3) {      // (nu <= 3: the one-workgroup form of the same kernel, TDC_RESIDENT_MAX_NU)
        // per (unit slice c, wave): 4 KiB of bias registers + one 1 KiB fragment per k-step (rnn_cluster.hip)
        g->cl_dir[0] = (size_t)nu * 4 * (4096 + (size_t)(g->kh + 2 * kx0) * 1024);
        g->cl_dir[1] = (size_t)nu * 4 * (4096 + (size_t)(g->kh + cfg->dirs * g->kh) * 1024);
    }
    for (int l = 0; l < cfg->layers; ++l) {
        g->cl_off[l] = off;
        off += g->cl_dir[l == 0 ? 0 : 1] * cfg->dirs;
    }
    g->packed_bytes = off;
    return TD_OK;
}


// Geometry of the MFMA-ordered recurrent weights and activation fragments (host + device).
//
// Activations travel between layers as MFMA B-operand fragments of v_mfma_f32_32x32x16_bf16:
// one fragment = 64 lanes x 8 bf16 (1 KiB) = a (16 k) x (32 batch) tile; lane (m = lane & 31,
// hf = lane >> 5), element j holds batch column m and
//     hidden unit  16*ks + 8*(j >> 2) + 4*hf + (j & 3)        (the order the accumulator leaves it in)
// for recurrent / inter-layer fragments, or input channel 16*kx + 8*hf + j for the first layer.
// The weights are packed once into A-operand fragments with the matching k permutation, so the
// new hidden state goes from the accumulator registers straight back into the next MFMA.
#pragma once
#include <stddef.h>
#include <stdint.h>

#define TD_FRAG_BYTES 1024
#define TD_STAGE_FRAGS 32                                   // fragments per LDS stage
#define TD_STAGE_BYTES (TD_STAGE_FRAGS * TD_FRAG_BYTES)     // 32 KiB
#define TD_BLOCK_WAVES 4                                    // waves per workgroup, 32 sequences each
#define TD_WAVE_BATCH 32

struct TdLayerGeom {
    int nu;        // 32-unit blocks of the padded hidden size
    int ng;        // weight fragments per k-step: 4 (LSTM i,f,g,o) or 3 (GRU r,z,n)
    int kh;        // recurrent k-steps = 2*nu
    int kx;        // input k-steps: ceil(c_in/16) for layer 0, din*kh above
    int xfr;       // input fragments held per step: 2*kx for layer 0 (bf16 hi + lo), kx above
    int fu;        // weight fragments per unit block
    int nf;        // weight fragments per time step
    int ns;        // LDS stages per time step
    size_t frag_bytes;  // ns * TD_STAGE_BYTES
    size_t dir_bytes;   // = frag_bytes (one direction of one layer)
};

static inline int td_round_nu(int hidden) {
    int nu = (hidden + 31) / 32;
    // up to 8 blocks (H <= 256): rnn_layer.h, one wave holds a batch block's whole state; 12..32 blocks
    // (H <= 1024): rnn_wide.hip, the four waves of a workgroup split the unit blocks
    static const int ok[] = {1, 2, 3, 4, 6, 8, 12, 16, 24, 32};
    for (int i = 0; i < 10; ++i)
        if (nu <= ok[i]) return ok[i];
    return -1;
}

static inline TdLayerGeom td_layer_geom(int nu, int gru, int layer0, int kx_in /* ceil(c_in/16) or din */) {
    TdLayerGeom g;
    g.nu = nu;
    g.ng = gru ? 3 : 4;
    g.kh = 2 * nu;
    g.kx = layer0 ? kx_in : kx_in * g.kh;
    g.xfr = layer0 ? 2 * g.kx : g.kx;
    g.fu = 4 + (g.kh + g.xfr) * g.ng;   // 4 bias fragments lead each unit block
    g.nf = nu * g.fu;
    g.ns = (g.nf + TD_STAGE_FRAGS - 1) / TD_STAGE_FRAGS;
    g.frag_bytes = (size_t)g.ns * TD_STAGE_BYTES;
    g.dir_bytes = g.frag_bytes;
    return g;
}

// Launch parameters of the recurrent-layer kernel (rnn_layer.h), shared with the host-side dispatch.
#pragma once
#include <stddef.h>

struct TdLayerParams {
    const char *wpacked;  // packed weights of this layer, direction 0
    size_t dir_stride;    // bytes to direction 1
    const char *xin;      // input fragments  [l][bblk][in_frags][64][8] bf16
    char *out;            // output fragments [l][bblk][out_frags][64][8] bf16
    int in_frags, out_frags;
    int L, nbb, dirs;
    // training only (SAVE kernels): gate activations and state for the backward pass,
    // [dir][l][bblk][u][5][64 lanes][16] fp32 in accumulator layout: LSTM i,f,g,o,c ; GRU r,z,n,(W_hn h + b_hn),h
    float *save;
    size_t save_dir_stride;  // floats
    // wide layers (rnn_wide.hip) only: recurrent state between steps, [dir][bblk][u][64 lanes][16] fp32
    float *state;
    size_t state_dir_stride;  // floats
    // HEAD kernels (the sampler's top layer) only: per (step, batch block, direction, lane) hp_stride floats: the partial
    // projection's accumulator registers (8, or 16 with more than 16 output channels), then sum h, sum h^2, 0, 0
    float *hpart;
    int hp_stride;            // 12 or 20
};

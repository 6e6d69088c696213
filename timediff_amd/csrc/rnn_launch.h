// Dispatch from runtime (hidden block count, input k-steps, layer kind, cell) to the instantiated
// recurrent-layer kernels.  Every (hidden block count, cell, training) combination is its own translation
// unit (rnn_layer_inst.hip compiled with -DTD_INST_NU/-DTD_INST_GRU/-DTD_INST_SAVE): the kernels are fully
// unrolled over one time step, and one file with all 96 of them takes the better part of an hour to compile.
#pragma once
#include "common.h"
#include "rnn_params.h"

int td_launch_layer_nu1_g0_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu2_g0_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu3_g0_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu4_g0_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu6_g0_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu8_g0_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu1_g1_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu2_g1_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu3_g1_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu4_g1_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu6_g1_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu8_g1_s0(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu1_g0_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu2_g0_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu3_g0_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu4_g0_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu6_g0_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu8_g0_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu1_g1_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu2_g1_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu3_g1_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu4_g1_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu6_g1_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);
int td_launch_layer_nu8_g1_s1(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream);

// head: the fused-head variant (inference kernels of the layers above the first only; rnn_layer.h)
static inline int td_launch_layer_dispatch(int gru, int save, int nu, const TdLayerParams &p, bool layer0, int kxsel,
                                           hipStream_t stream, bool head = false) {
    if (!gru && !save) switch (nu) {
        case 1: return td_launch_layer_nu1_g0_s0(p, layer0, kxsel, head, stream);
        case 2: return td_launch_layer_nu2_g0_s0(p, layer0, kxsel, head, stream);
        case 3: return td_launch_layer_nu3_g0_s0(p, layer0, kxsel, head, stream);
        case 4: return td_launch_layer_nu4_g0_s0(p, layer0, kxsel, head, stream);
        case 6: return td_launch_layer_nu6_g0_s0(p, layer0, kxsel, head, stream);
        case 8: return td_launch_layer_nu8_g0_s0(p, layer0, kxsel, head, stream);
    }
    if (gru && !save) switch (nu) {
        case 1: return td_launch_layer_nu1_g1_s0(p, layer0, kxsel, head, stream);
        case 2: return td_launch_layer_nu2_g1_s0(p, layer0, kxsel, head, stream);
        case 3: return td_launch_layer_nu3_g1_s0(p, layer0, kxsel, head, stream);
        case 4: return td_launch_layer_nu4_g1_s0(p, layer0, kxsel, head, stream);
        case 6: return td_launch_layer_nu6_g1_s0(p, layer0, kxsel, head, stream);
        case 8: return td_launch_layer_nu8_g1_s0(p, layer0, kxsel, head, stream);
    }
    if (!gru && save) switch (nu) {
        case 1: return td_launch_layer_nu1_g0_s1(p, layer0, kxsel, head, stream);
        case 2: return td_launch_layer_nu2_g0_s1(p, layer0, kxsel, head, stream);
        case 3: return td_launch_layer_nu3_g0_s1(p, layer0, kxsel, head, stream);
        case 4: return td_launch_layer_nu4_g0_s1(p, layer0, kxsel, head, stream);
        case 6: return td_launch_layer_nu6_g0_s1(p, layer0, kxsel, head, stream);
        case 8: return td_launch_layer_nu8_g0_s1(p, layer0, kxsel, head, stream);
    }
    if (gru && save) switch (nu) {
        case 1: return td_launch_layer_nu1_g1_s1(p, layer0, kxsel, head, stream);
        case 2: return td_launch_layer_nu2_g1_s1(p, layer0, kxsel, head, stream);
        case 3: return td_launch_layer_nu3_g1_s1(p, layer0, kxsel, head, stream);
        case 4: return td_launch_layer_nu4_g1_s1(p, layer0, kxsel, head, stream);
        case 6: return td_launch_layer_nu6_g1_s1(p, layer0, kxsel, head, stream);
        case 8: return td_launch_layer_nu8_g1_s1(p, layer0, kxsel, head, stream);
    }
    td_set_error("unsupported hidden block count %d", nu);
    return TD_ERR_UNSUPPORTED;
}

// Dispatch from runtime (hidden block count, input k-steps, layer kind, cell) to the instantiated
// recurrent-layer kernels.  Included by rnn.hip (inference kernels) and rnn_train.hip (SAVE kernels).
#pragma once
#include "rnn_layer.h"

template <int NU, int KX, bool L0, bool GRU, bool SAVE>
static int td_launch_layer(const TdLayerParams &p, hipStream_t stream) {
    const int tiles = td_cdiv(p.nbb, TD_BLOCK_WAVES);
    hipLaunchKernelGGL((td_rnn_layer_kernel<NU, KX, L0, GRU, SAVE>), dim3(tiles * p.dirs), dim3(256), 0, stream, p);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

template <int NU, bool GRU, bool SAVE>
static int td_launch_layer_nu(const TdLayerParams &p, bool layer0, int kxsel, hipStream_t stream) {
    if (layer0) {
        if (kxsel == 1) return td_launch_layer<NU, 1, true, GRU, SAVE>(p, stream);
        return td_launch_layer<NU, 2, true, GRU, SAVE>(p, stream);
    }
    if (kxsel == 1) return td_launch_layer<NU, 2 * NU, false, GRU, SAVE>(p, stream);
    return td_launch_layer<NU, 4 * NU, false, GRU, SAVE>(p, stream);
}

template <bool GRU, bool SAVE>
static int td_launch_layer_any(int nu, const TdLayerParams &p, bool layer0, int kxsel, hipStream_t stream) {
    switch (nu) {
        case 1: return td_launch_layer_nu<1, GRU, SAVE>(p, layer0, kxsel, stream);
        case 2: return td_launch_layer_nu<2, GRU, SAVE>(p, layer0, kxsel, stream);
        case 3: return td_launch_layer_nu<3, GRU, SAVE>(p, layer0, kxsel, stream);
        case 4: return td_launch_layer_nu<4, GRU, SAVE>(p, layer0, kxsel, stream);
        case 6: return td_launch_layer_nu<6, GRU, SAVE>(p, layer0, kxsel, stream);
        case 8: return td_launch_layer_nu<8, GRU, SAVE>(p, layer0, kxsel, stream);
    }
    td_set_error("unsupported hidden block count %d", nu);
    return TD_ERR_UNSUPPORTED;
}


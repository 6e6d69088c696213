// One translation unit per (hidden block count, cell, training): see rnn_launch.h.
//   hipcc ... -DTD_INST_NU=8 -DTD_INST_GRU=0 -DTD_INST_SAVE=0 -c rnn_layer_inst.hip -o build/layer_nu8_g0_s0.o
#include "rnn_layer.h"

#define TD_CAT2(a, b, c, d, e, f) a##b##c##d##e##f
#define TD_CAT(a, b, c, d, e, f) TD_CAT2(a, b, c, d, e, f)
#define TD_INST_NAME TD_CAT(td_launch_layer_nu, TD_INST_NU, _g, TD_INST_GRU, _s, TD_INST_SAVE)

template <int NU, int KX, bool L0, bool GRU, bool SAVE>
static int td_launch_layer(const TdLayerParams &p, hipStream_t stream) {
    const int tiles = td_cdiv(p.nbb, TD_BLOCK_WAVES);
    hipLaunchKernelGGL((td_rnn_layer_kernel<NU, KX, L0, GRU, SAVE>), dim3(tiles * p.dirs), dim3(256), 0, stream, p);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

int TD_INST_NAME(const TdLayerParams &p, bool layer0, int kxsel, hipStream_t stream) {
    constexpr int NU = TD_INST_NU;
    constexpr bool GRU = TD_INST_GRU != 0, SAVE = TD_INST_SAVE != 0;
    if (layer0) {
        if (kxsel == 1) return td_launch_layer<NU, 1, true, GRU, SAVE>(p, stream);
        return td_launch_layer<NU, 2, true, GRU, SAVE>(p, stream);
    }
    if (kxsel == 1) return td_launch_layer<NU, 2 * NU, false, GRU, SAVE>(p, stream);
    return td_launch_layer<NU, 4 * NU, false, GRU, SAVE>(p, stream);
}

// One translation unit per (hidden block count, cell, training): see rnn_launch.h.
//   hipcc ... -DTD_INST_NU=8 -DTD_INST_GRU=0 -DTD_INST_SAVE=0 -c rnn_layer_inst.hip -o build/layer_nu8_g0_s0.o
#include "rnn_layer.h"

#define TD_CAT2(a, b, c, d, e, f) a##b##c##d##e##f
#define TD_CAT(a, b, c, d, e, f) TD_CAT2(a, b, c, d, e, f)
#define TD_INST_NAME TD_CAT(td_launch_layer_nu, TD_INST_NU, _g, TD_INST_GRU, _s, TD_INST_SAVE)

// Invariants of the compile-time schedule the kernel's counted waits rely on (checked per instantiation).
template <int NU, int KX, bool L0, bool GRU, bool SAVE, bool HEAD>
constexpr bool td_sched_ok() {
    using S = TdSched<NU, KX, L0, GRU, SAVE, HEAD>;
    using T = TdLayerTraits<NU, KX, L0, GRU, HEAD>;
    // every block after the first finishes all gate micro-ops of its predecessor inside its own slots, in order
    for (int u = 1; u < NU; ++u) {
        int q = 0;
        for (int f = 0; f < T::FU; ++f) {
            const int F = u * T::FU + f;
            if (S::tab.qbeg[F] != q || S::tab.qend[F] < q) return false;
            q = S::tab.qend[F];
        }
        if (q != S::NQ) return false;
    }
    for (int f = 0; f < T::FU; ++f)
        if (S::tab.qbeg[f] != S::tab.qend[f]) return false;   // block 0 carries none
    if (HEAD) {   // the head slots finish the last block's micro-ops, in order, before the last four of them
        int q = T::HF > 4 ? 0 : S::NQ;   // a single unit block: the kernel runs its gate math in front of the head slots
        for (int f = 0; f < T::HF; ++f) {
            const int F = T::NFB + f;
            if (S::tab.qbeg[F] != q || S::tab.qend[F] < q) return false;
            q = S::tab.qend[F];
            if (f >= T::HF - 4 && S::tab.qbeg[F] != S::NQ) return false;
        }
        if (q != S::NQ) return false;
    }
    // every ring stage gets exactly PIECES pieces per wave per step (the first APD reads go out in the step's prime)
    int pieces = 0;
    for (int R = 0; R < T::NF; ++R)
        for (int i = 0; i < T::PIECES; ++i) pieces += S::piece_pos(R / TD_RING_FRAGS, i) == R % TD_RING_FRAGS;
    if (pieces != T::PIECES * T::NS) return false;
    // every input fragment is replaced exactly once per step
    int xd = 0;
    for (int F = 0; F < T::NF; ++F) xd += S::xdma_of(F) >= 0;
    if (xd != (L0 ? 0 : KX)) return false;
    for (int j = 0; j < T::NS; ++j)
        if (S::tab.waitn[j] < 0 || S::tab.waitn[j] > 60) return false;   // vmcnt is a 6-bit field
    return true;
}

template <int NU, int KX, bool L0, bool GRU, bool SAVE, bool HEAD = false>
static int td_launch_layer(const TdLayerParams &p, hipStream_t stream) {
    static_assert(td_sched_ok<NU, KX, L0, GRU, SAVE, HEAD>(), "recurrent-layer schedule tables are inconsistent");
    const int tiles = td_cdiv(p.nbb, TD_BLOCK_WAVES);
    hipLaunchKernelGGL((td_rnn_layer_kernel<NU, KX, L0, GRU, SAVE, HEAD>), dim3(tiles * p.dirs), dim3(256), 0, stream, p);
    TD_LAUNCH_CHECK();
    return TD_OK;
}

int TD_INST_NAME(const TdLayerParams &p, bool layer0, int kxsel, bool head, hipStream_t stream) {
    constexpr int NU = TD_INST_NU;
    constexpr bool GRU = TD_INST_GRU != 0, SAVE = TD_INST_SAVE != 0;
    if (head) {
        if constexpr (!SAVE) {
            if (!layer0) {
                if (kxsel == 1) return td_launch_layer<NU, 2 * NU, false, GRU, false, true>(p, stream);
                return td_launch_layer<NU, 4 * NU, false, GRU, false, true>(p, stream);
            }
        }
        td_set_error("the fused-head layer kernel exists for inference launches of the layers above the first only");
        return TD_ERR_UNSUPPORTED;
    }
    if (layer0) {
        if (kxsel == 1) return td_launch_layer<NU, 1, true, GRU, SAVE>(p, stream);
        return td_launch_layer<NU, 2, true, GRU, SAVE>(p, stream);
    }
    if (kxsel == 1) return td_launch_layer<NU, 2 * NU, false, GRU, SAVE>(p, stream);
    return td_launch_layer<NU, 4 * NU, false, GRU, SAVE>(p, stream);
}

"""Gradients under the cluster BPTT sweep against the single-workgroup sweep on the same saved values (debug aid):
python3 tools/bptt_probe.py 600 1024 2048"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
sys.path.insert(0, "tests")
from timediff_amd import _lib
from test_gpu_parity import cu, den_kernels, _rel
from test_gpu_wide import _inputs, random_state

dev = torch.device("cuda:0")
L = int(sys.argv[1])
for B in [int(a) for a in sys.argv[2:]]:
    sd = random_state(256, 14, 3, 2, 0, seed=B)
    den = den_kernels(sd, dev)
    x, t, rng = _inputs(B, 14, L, 4, seed=B)
    gy = (rng.standard_normal((B, 14, L)) / (B * L)).astype(np.float32)
    film = den.time_film(cu(t, dev))
    grads = {}
    for mode in (_lib.LAYER_KERNEL_CLUSTER, _lib.LAYER_KERNEL_UNITS):
        _lib.check(_lib.lib().td_set_layer_kernel(mode))
        den.forward_train(cu(x, dev), film)
        g_layer, g_misc, _ = den.backward(cu(gy, dev), film)
        grads[mode] = [g.cpu().numpy().copy() for g in g_layer]
    errs = [(k.split(".")[-1], round(_rel(a, b), 4)) for k, a, b in zip(den._lkeys, grads[3], grads[2])]
    print(f"B={B} L={L}", errs, flush=True)

"""Denoiser forward at the eICU shape under each layer-kernel family (not part of the product):
python3 tools/family_probe.py 512 2048 16384"""
import sys, time
import torch
sys.path.insert(0, ".")
from timediff_amd import _lib
from timediff_amd.ops import DenoiserKernels
from tools.perf_probe import make_sd

dev = torch.device("cuda:0")
L, C = 272, 14
den = DenoiserKernels(make_sd())
names = {0: "auto", 1: "batch", 2: "units", 3: "cluster"}
for B in [int(a) for a in sys.argv[1:]] or [512]:
    x = torch.randn(B, C, L, device=dev)
    t = torch.randint(0, 1000, (B,), device=dev)
    film = den.time_film(t)
    ref = None
    for mode in (3, 1, 2, 0):
        _lib.check(_lib.lib().td_set_layer_kernel(mode))
        y = den.forward(x, film=film)
        torch.cuda.synchronize()
        n = 3
        t0 = time.time()
        for _ in range(n):
            y = den.forward(x, film=film)
        torch.cuda.synchronize()
        dt = (time.time() - t0) / n
        if ref is None:
            ref = y
        print(f"B={B} {names[mode]:8s} fwd {dt*1e3:8.2f} ms  {2.017e9*B/dt/1e12:7.1f} TFLOP/s  max|y - y_cluster| = {(y-ref).abs().max().item():.2e} finite={bool(torch.isfinite(y).all())}", flush=True)

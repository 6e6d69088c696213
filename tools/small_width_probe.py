"""Denoiser forward (inference and saving) at the small reference widths under each layer-kernel family:
python3 tools/small_width_probe.py 256 2048 16384"""
import sys, time
import torch
sys.path.insert(0, ".")
from bench import init_state
from timediff_amd import _lib
from timediff_amd.ops import DenoiserKernels

dev = torch.device("cuda:0")
names = {0: "auto", 1: "batch", 2: "units", 3: "cluster"}
for tag, C_io, L, H in (("mimiciii", 23, 24, 92), ("hirid", 9, 100, 36)):
    den = DenoiserKernels({k: v.to(dev) for k, v in init_state(C_io, H, 3, 2, seed=2023).items()})
    for B in [int(a) for a in sys.argv[1:]] or [256]:
        x = torch.randn(B, C_io, L, device=dev)
        t = torch.randint(0, 1000, (B,), device=dev)
        film = den.time_film(t)
        ref = None
        for mode in (3, 1, 2):
            _lib.check(_lib.lib().td_set_layer_kernel(mode))
            if _lib.lib().td_rnn_layer_kernel_choice(__import__("ctypes").byref(den.cfg), B, L) != mode:
                continue
            out = {}
            for kind, fn in (("infer", lambda: den.forward(x, film=film)), ("save", lambda: den.forward_train(x, film))):
                y = fn()
                torch.cuda.synchronize()
                n = 5
                t0 = time.time()
                for _ in range(n):
                    y = fn()
                torch.cuda.synchronize()
                out[kind] = (time.time() - t0) / n * 1e3
            if ref is None:
                ref = y
            print(f"{tag} H={H} L={L} B={B} {names[mode]:8s} infer {out['infer']:8.3f} ms  save {out['save']:8.3f} ms  "
                  f"max|y - y_first| = {(y - ref).abs().max().item():.2e} finite={bool(torch.isfinite(y).all())}", flush=True)
        _lib.check(_lib.lib().td_set_layer_kernel(0))

"""Forward time under the forced cluster kernel (debug variants of the library via TIMEDIFF_HIP_LIB): python3 tools/cluster_probe.py 16384"""
import sys, time
import torch
sys.path.insert(0, ".")
from timediff_amd import _lib
from timediff_amd.ops import DenoiserKernels
from tools.perf_probe import make_sd
dev = torch.device("cuda:0")
den = DenoiserKernels(make_sd())
_lib.check(_lib.lib().td_set_layer_kernel(3))
for B in [int(a) for a in sys.argv[1:]]:
    x = torch.randn(B, 14, 272, device=dev); t = torch.randint(0, 1000, (B,), device=dev); film = den.time_film(t)
    y = den.forward(x, film=film); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(3): y = den.forward(x, film=film)
    torch.cuda.synchronize()
    print(f"B={B} cluster fwd {(time.time()-t0)/3*1e3:.2f} ms", flush=True)

"""Per-step phase times of one wave of the single-block cluster forward kernel (library built with -DTDC_TRACE):
TIMEDIFF_HIP_LIB=tools/kbench/bin/libtrace.so python3 tools/cluster_trace.py [batch]"""
import ctypes as C
import sys

import numpy as np
import torch
sys.path.insert(0, ".")
from timediff_amd import _lib
from timediff_amd.ops import DenoiserKernels
from tools.perf_probe import make_sd

dev = torch.device("cuda:0")
den = DenoiserKernels(make_sd())
lib = _lib.lib()
_lib.check(lib.td_set_layer_kernel(3))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
L = 272
x = torch.randn(B, 14, L, device=dev)
t = torch.randint(0, 1000, (B,), device=dev)
film = den.time_film(t)
for _ in range(3):
    y = den.forward(x, film=film)
torch.cuda.synchronize()
buf = np.zeros(4096 * 8, dtype=np.uint64)
lib.td_debug_cluster_trace.restype = C.c_int
assert lib.td_debug_cluster_trace(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size)) == 0
tr = buf.reshape(4096, 8)[:L].astype(np.int64)
ns = 10.0      # wall_clock64 ticks of 10 ns
seg = np.diff(tr[:, :6], axis=1) * ns
names = ["issue+x-mfma", "poll", "lds+barrier", "h-mfma", "finish"]
body = slice(8, L - 2)
print("step period (ns): median", np.median(np.diff(tr[:, 0])[body]) * ns)
for i, n in enumerate(names):
    print(f"  {n:14s} median {np.median(seg[body, i]):7.0f}  p90 {np.percentile(seg[body, i], 90):7.0f}")
tail = (tr[1:, 0] - tr[:-1, 5]) * ns
print(f"  {'tail->next top':14s} median {np.median(tail[body]):7.0f}")
print("  spins: median", np.median(tr[body, 6]), "max", tr[body, 6].max())

# ---- BPTT sweep of the top layer (the last launch leaves its stamps) ----
if hasattr(lib, "td_debug_bptt_trace") or True:
    xg = torch.randn(B, 14, L, device=dev)
    gy = torch.randn(B, 14, L, device=dev) / (B * L)
    film_b = den.time_film(t)
    for _ in range(2):
        den.forward_train(xg, film_b)
        den.backward(gy, film_b)
    torch.cuda.synchronize()
    buf = np.zeros(4096 * 8, dtype=np.uint64)
    lib.td_debug_bptt_trace.restype = C.c_int
    assert lib.td_debug_bptt_trace(buf.ctypes.data_as(C.c_void_p), C.c_size_t(buf.size)) == 0
    tr = buf.reshape(4096, 8)[:L].astype(np.int64)
    body = slice(8, L - 3)
    print("BPTT step period (ns): median", np.median(np.diff(tr[:, 0])[body]) * ns)
    seg = np.diff(tr[:, :6], axis=1) * ns
    for i, n in enumerate(["poll (hop)", "gates+stores", "barrier", "prefetch+mfma", "send"]):
        print(f"  {n:14s} median {np.median(seg[body, i]):7.0f}  p90 {np.percentile(seg[body, i], 90):7.0f}")
    tail = (tr[1:, 0] - tr[:-1, 5]) * ns
    print(f"  {'send->next top':14s} median {np.median(tail[body]):7.0f}")
    print("  spins: median", np.median(tr[body, 6]), "max", tr[body, 6].max())

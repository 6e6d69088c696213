"""Two trainers from identical state and seeds, launched kernel by kernel: which flat-gradient entries differ run to run?"""
import os, sys
import numpy as np
import torch
sys.path.insert(0, "."); sys.path.insert(0, "tests")
os.environ["TIMEDIFF_TRAIN_GRAPH"] = "0"
from conftest import load_golden, sub
from test_gpu_train import _build
import pathlib
dev = torch.device("cuda:0")
g = load_golden("train_steps_eicu")
L, H, N, T, B, ga = [int(v) for v in g["cfg"]]
data = torch.from_numpy(g["data"].reshape(-1, *g["data"].shape[2:]).copy())
rows = torch.randint(0, data.shape[0], (6, 2 * B), generator=torch.Generator().manual_seed(5))
ts = torch.randint(0, T, (6, 2 * B), generator=torch.Generator().manual_seed(6))
tmp = pathlib.Path("/tmp")
names = None
grads = []
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 4):
    et = _build(g, dev, ga, 1e-3, tmp, T, L, H, N, True)
    if names is None:
        names, off = [], 0
        for n, p in et.model.named_parameters():
            names.append((n, off, off + p.numel())); off += p.numel()
    et.dl = iter([data[r].clone() for r in rows.reshape(12, B)])
    t_dev = torch.empty(2 * B, dtype=torch.int64, device=dev)
    inner = et.model.forward
    et.model.forward = lambda sample, inner=inner, t_dev=t_dev: inner(sample, t=t_dev)
    torch.manual_seed(77)
    snap = []
    for step in range(3):
        t_dev.copy_(ts[step])
        et.train_step()
        snap.append(et.exp_avg.clone())
    torch.cuda.synchronize()
    grads.append(snap)
for rep in range(1, len(grads)):
    for step in range(3):
        d = (grads[rep][step] != grads[0][step]).nonzero().flatten().cpu().numpy()
        if len(d):
            where = {}
            for i in d:
                for n, a, b in names:
                    if a <= i < b:
                        where.setdefault(n, []).append(int(i - a))
            print(f"rep {rep} step {step}: {len(d)} entries differ:", {k: (len(v), v[:8]) for k, v in where.items()})
            break
    else:
        print(f"rep {rep}: identical")

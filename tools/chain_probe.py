#!/usr/bin/env python3
"""One sampler chain at a reference shape (for timelines): python3 tools/chain_probe.py <mimiciii|hirid|eicu> <batch> [T]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import EICU, init_state  # noqa: E402
from tools.config_bench import SHAPES  # noqa: E402


def main():
    from timediff_amd.ops import DenoiserKernels, DiffusionKernels, SamplerKernels
    from timediff_amd.schedule import make_tables
    name, B = sys.argv[1], int(sys.argv[2])
    T = int(sys.argv[3]) if len(sys.argv) > 3 else 64
    cols, L, H = dict(SHAPES, eicu=(EICU, 272, 256))[name]
    dev = torch.device("cuda:0")
    C_io, C = len(cols["num"]) + sum(cols["K"]), len(cols["num"]) + len(cols["cat"])
    den = DenoiserKernels({k: v.to(dev) for k, v in init_state(C_io, H, 3, 2, seed=2023).items()})
    diff = DiffusionKernels(cols["num"], cols["cat"], cols["K"], L, T, make_tables(T, "cosine"), C, dev)
    sampler = SamplerKernels(den, diff)
    sampler.sample(B, seed=1)
    torch.cuda.synchronize()
    t0 = time.time()
    sampler.sample(B, seed=2)
    torch.cuda.synchronize()
    print(f"{name} B={B}: {(time.time() - t0) / T * 1e3:.4f} ms per reverse step", flush=True)


if __name__ == "__main__":
    main()

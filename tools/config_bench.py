#!/usr/bin/env python3
"""Sampler and training rates at the shapes of BASELINE.json's configs other than the headline one (which bench.py
measures): MIMIC-III-shaped (configs[0]) and HiRID-shaped at the reference's own width, on one MI355X.
    python3 tools/config_bench.py            -> one JSON line per shape
Sampler: a T = 64 chain timed and scaled to the reference's T = 1000 (the per-step cost does not depend on t)."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import init_state, train_probe  # noqa: E402

SHAPES = {
    # name: (columns, L, H): etdiff_train.py:24-29, :239 (hidden = 4 x denoiser width for these datasets)
    "mimiciii": (dict(num=[0, 2, 4, 6, 8, 10, 12], cat=[1, 3, 5, 7, 9, 11, 13, 14], K=[2] * 8), 24, 92),
    "hirid": (dict(num=[0, 1, 2, 3, 4, 5, 6], cat=[7], K=[2]), 100, 36),
}


def main():
    from timediff_amd.ops import DenoiserKernels, DiffusionKernels, SamplerKernels
    from timediff_amd.schedule import make_tables
    dev = torch.device("cuda:0")
    T, N, D = 64, 3, 2
    for name, (cols, L, H) in SHAPES.items():
        C_io = len(cols["num"]) + sum(cols["K"])
        C = len(cols["num"]) + len(cols["cat"])
        den = DenoiserKernels({k: v.to(dev) for k, v in init_state(C_io, H, N, D, seed=2023).items()})
        diff = DiffusionKernels(cols["num"], cols["cat"], cols["K"], L, T, make_tables(T, "cosine"), C, dev)
        sampler = SamplerKernels(den, diff)
        out = {"shape": name, "C_io": C_io, "L": L, "hidden": H, "sampler": {}}
        for B in (256, 16384, 131072):
            sampler.sample(B, seed=1)
            torch.cuda.synchronize()
            t0 = time.time()
            y = sampler.sample(B, seed=2)
            torch.cuda.synchronize()
            step = (time.time() - t0) / T
            assert torch.isfinite(y).all()
            out["sampler"][str(B)] = {"reverse_step_ms": step * 1e3, "sequences_per_s_at_T1000": B / (step * 1000)}
        del sampler, diff, den
        torch.cuda.empty_cache()
        r = train_probe(dev, cols, L, H, N, 1000, 256, 1, 200, 10)      # a 1 ms step: enough of them that the capture and the first replays do not weigh
        out["train_batch256"] = {"optimizer_steps_per_s": r["value"], "ms_per_step": r["ms_per_step"]}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()

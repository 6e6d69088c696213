#!/usr/bin/env python3
"""Host-side profile (cProfile) of the training step at the reference's batch: where the ~2 ms per step beyond the GPU
time go.  python3 tools/train_hostprof.py [steps]"""
import cProfile
import os
import pstats
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
holder = {}
orig = bench.train_probe


def run():
    return orig(dev, bench.EICU, 272, 256, 3, 1000, 32, 2, steps, 5)


pr = cProfile.Profile()
t0 = time.time()
pr.enable()
res = run()
pr.disable()
print(res["value"], "steps/s", res["ms_per_step"], "ms/step")
pstats.Stats(pr).sort_stats("tottime").print_stats(28)

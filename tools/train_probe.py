#!/usr/bin/env python3
"""One training probe of bench.py on its own (for rocprofv3 runs): python3 tools/train_probe.py --batch 32 --ga 2 --steps 20"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import EICU, train_probe  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--ga", type=int, default=2)
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--warmup", type=int, default=3)
ap.add_argument("--hidden", type=int, default=256)
ap.add_argument("--seq-len", type=int, default=272)
ap.add_argument("--shape", default="eicu", help="eicu (bench.py), or mimiciii / hirid at their reference widths (tools/config_bench.py); "
                                               "the latter set --hidden and --seq-len themselves")
ap.add_argument("--no-side-gemm", action="store_true", help="weight-gradient GEMMs on the caller's stream (td_set_side_gemm(0))")
ap.add_argument("--loop", action="store_true", help="gradient accumulation as a loop over micro-batches (the reference's form)")
a = ap.parse_args()
torch.cuda.set_device(0)
if a.no_side_gemm:
    from timediff_amd import _lib
    _lib.check(_lib.lib().td_set_side_gemm(0))
cols = EICU
if a.shape != "eicu":
    from tools.config_bench import SHAPES
    cols, a.seq_len, a.hidden = SHAPES[a.shape]
print(json.dumps(train_probe(torch.device("cuda", 0), cols, a.seq_len, a.hidden, 3, 1000, a.batch, a.ga, a.steps, a.warmup, fuse=not a.loop)))

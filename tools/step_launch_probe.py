#!/usr/bin/env python3
"""Training step rate with the unit-split family's per-step launches forced on / off (td_set_step_launch):
python3 tools/step_launch_probe.py <hidden> <batch> [<batch> ...]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import EICU, train_probe  # noqa: E402
from timediff_amd import _lib  # noqa: E402

H = int(sys.argv[1])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
for B in [int(a) for a in sys.argv[2:]]:
    row = {"hidden": H, "batch": B}
    for name, mode in (("persistent", _lib.STEP_LAUNCH_OFF), ("per_step", _lib.STEP_LAUNCH_ON)):
        _lib.check(_lib.lib().td_set_step_launch(mode))
        r = train_probe(dev, EICU, 272, H, 3, 1000, B, 1, 4, 2)
        row[name] = {"steps_per_s": round(r["value"], 2),
                     **{k.split(",")[0]: round(v["avg_launch_ms"], 2) for k, v in r["roofline"]["kernels"].items()}}
        torch.cuda.empty_cache()
    print(json.dumps(row), flush=True)
_lib.check(_lib.lib().td_set_step_launch(_lib.STEP_LAUNCH_AUTO))

"""BASELINE.json configs[3]: long-sequence HiRID-shaped workload (C_io = 9: 7 vitals + one binary label, hundreds
of time steps, wider hidden size) through the wide recurrent-layer kernel (rnn_wide.hip).

Times the denoiser forward (HIP events around each upper-layer launch, td_profile_*) and a short reverse chain:
    python3 tools/wide_bench.py [--hidden 512 1024] [--seq-len 512] [--batch 4096] [--timesteps 8]
Prints one JSON line per hidden size.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import denoiser_flops_per_seq, init_state  # noqa: E402

HIRID = dict(num=[0, 1, 2, 3, 4, 5, 6], cat=[7], K=[2])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--hidden", type=int, nargs="+", default=[512, 1024])
    ap.add_argument("--seq-len", type=int, default=512)
    ap.add_argument("--batch", type=int, default=4096)
    ap.add_argument("--timesteps", type=int, default=8)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--train-batch", type=int, default=0,
                    help="also time the training step (BASELINE.json configs[4]: wide-hidden BRNN at a large batch), "
                         "eICU-shaped, this many sequences per optimizer step on this GPU")
    args = ap.parse_args()
    from timediff_amd import _lib
    from timediff_amd.ops import DenoiserKernels, DiffusionKernels, SamplerKernels
    from timediff_amd.schedule import make_tables
    dev = torch.device("cuda:0")
    cols, L, B, T, N, D = HIRID, args.seq_len, args.batch, args.timesteps, 3, 2
    C_io = len(cols["num"]) + sum(cols["K"])
    lib = _lib.lib()
    for H in args.hidden:
        sd = init_state(C_io, H, N, D, seed=2023)
        den = DenoiserKernels({k: v.to(dev) for k, v in sd.items()})
        x = torch.randn(B, C_io, L, device=dev)
        film = den.time_film(torch.tensor([T - 1], device=dev))
        y = den.forward(x, film=film)
        torch.cuda.synchronize()
        _lib.check(lib.td_profile_enable(1))
        t0 = time.time()
        for _ in range(args.reps):
            y = den.forward(x, film=film)
        torch.cuda.synchronize()
        fwd_s = (time.time() - t0) / args.reps
        cnt, ms = C.c_int32(0), C.c_float(0)
        _lib.check(lib.td_profile_read(C.byref(cnt), C.byref(ms)))
        _lib.check(lib.td_profile_enable(0))
        layer_s = ms.value / max(cnt.value, 1) * 1e-3
        flops_layer = B * D * 2.0 * L * 4 * H * (D * H + H)
        diff = DiffusionKernels(cols["num"], cols["cat"], cols["K"], L, T, make_tables(T, "cosine"), 8, dev)
        sampler = SamplerKernels(den, diff)
        sampler.sample(B, seed=1)
        torch.cuda.synchronize()
        t0 = time.time()
        out = sampler.sample(B, seed=2)
        torch.cuda.synchronize()
        step_s = (time.time() - t0) / T
        assert torch.isfinite(out).all()
        print(json.dumps({
            "workload": f"HiRID-shaped: C_io={C_io}, L={L}, 3-layer bidirectional LSTM H={H}, batch {B}",
            "denoiser_forward_ms": fwd_s * 1e3,
            "denoiser_tflops": denoiser_flops_per_seq(C_io, H, N, D, L) * B / fwd_s / 1e12,
            "upper_layer_launch_ms": layer_s * 1e3, "upper_layer_tflops": flops_layer / layer_s / 1e12,
            "upper_layer_frac_of_bf16_peak": flops_layer / layer_s / 1e12 / 2500.0,
            "reverse_step_ms": step_s * 1e3, "sequences_per_s_at_T1000": B / (step_s * 1000),
        }), flush=True)
        del den, sampler, diff, x, y, out
        torch.cuda.empty_cache()
        if args.train_batch:
            from bench import EICU, train_probe
            r = train_probe(dev, EICU, 272, H, N, 1000, args.train_batch, 1, 3, 1)
            r["denoiser_tflops_fwd_bwd"] = 3 * denoiser_flops_per_seq(14, H, N, D, 272) * r["sequences_per_s"] / 1e12
            print(json.dumps({"train": r}), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()

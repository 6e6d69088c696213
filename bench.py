#!/usr/bin/env python3
"""Headline benchmark: synthetic patient sequences / second through the FULL T = 1000 reverse chain.

Workload (BASELINE.json configs[1]): eICU-shaped ETDiff -- 4 continuous + 5 binary channels
(C_io = 14), L = 272, 3-layer bidirectional LSTM denoiser with H = 256, random-init weights
(seed 2023), bf16 MFMA denoiser with fp32 state, in-kernel Philox noise.  One "step" = one complete
chain (prior draw -> T x (denoiser + reverse update) -> decode) for `--batch` sequences per GPU, replayed
as a captured hipGraph.  Patients are sharded across ranks (disjoint Philox patient ids), no data-path
collective: weak scaling.

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A chain of 16 384 sequences takes ~30 s, so `--steps` / `--warmup` are CAPS under a wall budget (`--budget-s`, default
240 s for warm-up + timed chains together): warm-up is a short-T pass through every kernel plus at most one full
chain, and timed chains stop when the next one would not fit.  The JSON line reports the `steps` and `warmup`
actually executed (and the requested caps beside them); every rank runs the same number of chains.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

EICU = dict(num=[0, 2, 4, 6], cat=[1, 3, 5, 7, 8], K=[2, 2, 2, 2, 2])
MFMA_BF16_PEAK_TFLOPS = 2500.0      # dense bf16, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"


def init_state(C_io, H, N, D, seed):
    """Random-init weights of the architecture with torch's default nn.LSTM / nn.Linear initialisers."""
    g = torch.Generator().manual_seed(seed)

    def u(shape, k):
        return (torch.rand(*shape, generator=g) * 2 - 1) * k

    sd = {}
    k = 1.0 / np.sqrt(H)
    for l in range(N):
        for d in range(D):
            sfx = f"_l{l}" + ("_reverse" if d else "")
            In = C_io if l == 0 else D * H
            sd[f"rnn.rnn.weight_ih{sfx}"] = u((4 * H, In), k)
            sd[f"rnn.rnn.weight_hh{sfx}"] = u((4 * H, H), k)
            sd[f"rnn.rnn.bias_ih{sfx}"] = u((4 * H,), k)
            sd[f"rnn.rnn.bias_hh{sfx}"] = u((4 * H,), k)
    sd["time_emb.1.weight"] = u((256, 64), 1 / 8.0); sd["time_emb.1.bias"] = u((256,), 1 / 8.0)
    sd["time_emb.3.weight"] = u((256, 256), 1 / 16.0); sd["time_emb.3.bias"] = u((256,), 1 / 16.0)
    sd["rnn.time_mlp.1.weight"] = u((2 * D * H, 256), 1 / 16.0); sd["rnn.time_mlp.1.bias"] = u((2 * D * H,), 1 / 16.0)
    sd["rnn.norm.weight"] = torch.ones(D * H); sd["rnn.norm.bias"] = torch.zeros(D * H)
    kf = 1.0 / np.sqrt(D * H)
    sd["fc.weight"] = u((C_io, D * H), kf); sd["fc.bias"] = u((C_io,), kf)
    return sd


def denoiser_flops_per_seq(C_io, H, N, D, L):
    """BASELINE.md section 4 / SURVEY.md 8(d): algorithmic FLOPs of one denoiser forward for one sequence."""
    f = 0.0
    for l in range(N):
        In = C_io if l == 0 else D * H
        f += D * 2.0 * L * 4 * H * (In + H)
    f += 2.0 * L * D * H * C_io + 2.0 * (64 * 256 + 256 * 256 + 256 * 2 * D * H)
    return f


def cpu_baseline(sd, cols, L, T, tables, seconds_target=15.0):
    """The oracle (CPU restatement, fp32, OpenMP over sequences) timed on this box's host cores on a bounded
    sample of the same workload: a few reverse steps of a small batch, extrapolated to the T-step chain."""
    from oracle import oracle as orc
    threads = int(os.environ.get("OMP_NUM_THREADS", "0")) or min(16, os.cpu_count() or 1)
    den = orc.Denoiser({k: v.numpy() for k, v in sd.items()})
    od = orc.Diffusion(len(cols["num"]), cols["K"], L, T, tables=tables)
    B = 2 * threads
    rng = np.random.default_rng(0)
    z = rng.standard_normal((B, len(cols["num"]), L)).astype(np.float32)
    idx = rng.integers(0, 2, (B, len(cols["K"]), L)).astype(np.uint8)
    steps, t0 = 0, time.time()
    while True:
        t = np.full((B,), T - 1 - steps, np.int64)
        mo = den.forward(od.build_x_in(z, idx), t)
        z, idx = od.reverse_step(mo, z, idx, t, rng.standard_normal(z.shape).astype(np.float32),
                                 rng.random((B, sum(cols["K"]), L), dtype=np.float32))
        steps += 1
        el = time.time() - t0
        if el > seconds_target or steps >= 50:
            break
    per_step = el / steps
    return dict(value=B / (per_step * T), unit="sequences/s", cores=threads, kind="port",
                sample=f"{steps} reverse steps (denoiser + update) of {B} eICU-shaped sequences, fp32 oracle, "
                       f"{per_step:.2f} s/step, extrapolated to T={T}")


def parity_gates(dev, den, diff, sampler, sd, cols, L, T, tables, bench_batch, B=32):
    """SURVEY.md 8(d) "parity gates reported next to perf", at the bench's own architecture and weights: the HIP path
    against the CPU oracle (the checker, as in tests/) at t in {T-1, T/2, 10, 0} -- denoiser output, the reverse update's
    x0_hat / posterior mean / log theta, sampled class ids wherever the Gumbel-perturbed top-2 margin exceeds the
    tolerance, and the loss with its gradient.  The denoiser runs the kernel family the timed chains run: the small batch
    with the batch-block family forced (td_set_layer_kernel), and once more at the bench's own batch size under AUTO with
    the head fused into the top layer as in the chain, the oracle on a subset of the sequences."""
    from oracle import oracle as orc
    from timediff_amd import _lib
    lib = _lib.lib()
    names = {_lib.LAYER_KERNEL_BATCH: "batch-block (rnn_layer.h)", _lib.LAYER_KERNEL_UNITS: "unit-split (rnn_wide.hip)",
             _lib.LAYER_KERNEL_CLUSTER: "cluster (rnn_cluster.hip)"}
    timed_family = lib.td_rnn_layer_kernel_choice(C.byref(den.cfg), bench_batch, L)
    _lib.check(lib.td_set_layer_kernel(timed_family))
    try:
        gates = _parity_gates_body(dev, den, diff, sampler, sd, cols, L, T, tables, bench_batch, B, orc)
    finally:
        lib.td_set_layer_kernel(_lib.LAYER_KERNEL_AUTO)
        lib.td_set_head_fusion(_lib.HEAD_FUSION_AUTO)
    gates["kernel_family"] = names.get(timed_family, str(timed_family))
    return gates


def _parity_gates_body(dev, den, diff, sampler, sd, cols, L, T, tables, bench_batch, B, orc):
    from timediff_amd import _lib
    lib = _lib.lib()
    o16, o32 = orc.Denoiser({k: v.numpy() for k, v in sd.items()}, bf16=True), orc.Denoiser({k: v.numpy() for k, v in sd.items()})
    od = orc.Diffusion(len(cols["num"]), cols["K"], L, T, tables=tables)
    rng = np.random.default_rng(11)
    n_num, n_cat, sumk = len(cols["num"]), len(cols["K"]), sum(cols["K"])
    gates = {}
    for t in (T - 1, T // 2, min(10, T - 1), 0):
        z = rng.standard_normal((B, n_num, L)).astype(np.float32)
        idx = rng.integers(0, 2, (B, n_cat, L)).astype(np.uint8)
        noise = rng.standard_normal(z.shape).astype(np.float32)
        uni = rng.random((B, sumk, L), dtype=np.float32)
        tt = np.full((B,), t, np.int64)
        x_in = od.build_x_in(z, idx)
        mo = den.forward(torch.from_numpy(x_in).to(dev), torch.from_numpy(tt).to(dev))
        mo_h = mo.cpu().numpy()
        zg, ig = torch.from_numpy(z).to(dev), torch.from_numpy(idx).to(dev)
        x0, mean, lth = (a.cpu().numpy() for a in diff.reverse_step_terms(mo, zg, ig, t))
        z_ref, idx_ref, det = od.reverse_step(mo_h, z, idx, tt, noise, uni, details=True)
        diff.reverse_step_(mo, zg, ig, t, noise=torch.from_numpy(noise).to(dev), uniforms=torch.from_numpy(uni).to(dev))
        margin = od.gumbel_margin(det["log_theta"], uni)
        tol = 1e-3 if t <= 10 else 1e-5
        clear = margin > tol
        gates[f"t={t}"] = dict(
            model_out_vs_bf16_oracle=float(np.abs(mo_h - o16.forward(x_in, tt)).max()),
            model_out_vs_fp32_oracle=float(np.abs(mo_h - o32.forward(x_in, tt)).max()),
            x0_hat=float(np.abs(x0 - det["x0"]).max()), posterior_mean=float(np.abs(mean - det["mean"]).max()),
            log_theta=float(np.abs(lth - det["log_theta"]).max()), z_next=float(np.abs(zg.cpu().numpy() - z_ref).max()),
            class_ids_differing_above_margin=int((ig.cpu().numpy() != idx_ref)[clear].sum()),
            fraction_below_margin=float(1.0 - clear.mean()), margin=tol)
    t = rng.integers(0, T, B)
    t[0] = 0
    mo = rng.standard_normal((B, n_num + sumk, L)).astype(np.float32)
    eps = rng.standard_normal((B, n_num, L)).astype(np.float32)
    x0i = rng.integers(0, 2, (B, n_cat, L)).astype(np.uint8)
    xti = rng.integers(0, 2, (B, n_cat, L)).astype(np.uint8)
    ref = od.mixed_loss(mo, eps, x0i, xti, t, 0.8)
    loss, grad = diff.mixed_loss(torch.from_numpy(mo).to(dev), torch.from_numpy(eps).to(dev), torch.from_numpy(x0i).to(dev),
                                 torch.from_numpy(xti).to(dev), torch.from_numpy(t).to(dev), 0.8)
    gates["loss"] = dict(rel_err=float(abs(loss.item() - ref["loss"]) / abs(ref["loss"])),
                         grad_rel_err=float(np.abs(grad.cpu().numpy() - ref["grad"]).max() / np.abs(ref["grad"]).max()))
    gates["sample"] = f"{B} eICU-shaped sequences, L={L}, the bench's weights; tolerances as in tests/test_gpu_parity.py"
    # the timed launch shape itself: bench_batch sequences under AUTO, one FiLM row, head inside the top layer's kernel
    lib.td_set_layer_kernel(_lib.LAYER_KERNEL_AUTO)
    _lib.check(lib.td_set_head_fusion(_lib.HEAD_FUSION_ON))
    z, idx = diff.prior(bench_batch, seed=11)
    x_in = diff.build_input(z, idx)
    t = T - 1
    mo = den.forward(x_in, film=sampler._film[t:t + 1])
    rows = np.unique(np.concatenate([np.arange(0, 24), np.arange(0, bench_batch, max(bench_batch // 24, 1)),
                                     np.arange(bench_batch - 16, bench_batch)]))
    rows_t = torch.from_numpy(rows).to(dev)
    want = o16.forward(x_in[rows_t].cpu().numpy(), np.full(len(rows), t, np.int64))
    gates["timed_shape"] = dict(batch=bench_batch, head="fused into the top layer", rows_checked=int(len(rows)),
                                model_out_vs_bf16_oracle=float(np.abs(mo[rows_t].cpu().numpy() - want).max()))
    return gates


def _family_name(lib, et, seqs, L):
    """Which recurrent-layer kernel family AUTO takes for a launch of `seqs` sequences of the trainer's denoiser."""
    from timediff_amd import _lib
    den = et.model.model._kernels or et.model.model.kernels()
    fam = lib.td_rnn_layer_kernel_choice(C.byref(den.cfg), seqs, L)
    name = {_lib.LAYER_KERNEL_BATCH: "batch-block", _lib.LAYER_KERNEL_UNITS: "unit-split", _lib.LAYER_KERNEL_CLUSTER: "cluster"}.get(fam, str(fam))
    if lib.td_rnn_step_launch_choice(C.byref(den.cfg), seqs, L) == 1:
        name += ", one launch per time step"
    return name


def train_probe(dev, cols, L, H, N, T, batch, ga, steps, warmup, fuse=True):
    """Second half of BASELINE.json's metric: optimizer steps / second of the ETDiff training loop
    (et_diff.py:226-261: ga micro-batches of MixedDiffusion.forward + backward, clip, Adam, EMA) on synthetic
    eICU-shaped data, through the same module interface the reference's CLI uses."""
    from models.ETDiff.blocks import RNN
    from models.ETDiff.et_diff import ETDiff
    from models.ETDiff.mixed_diffusion import MixedDiffusion
    from timediff_amd.data import TensorSeriesDataset
    torch.manual_seed(2023)
    width = len(cols["num"]) + sum(cols["K"])
    n_ch = len(cols["num"]) + len(cols["cat"])
    rnn = RNN(input_channels=width, hidden_channels=H, output_channels=width, layers=N, model="lstm", bidirectional=True)
    diff = MixedDiffusion(rnn, seq_length=L, channels=width, numerical_features_indices=cols["num"],
                          categorical_features_indices=cols["cat"], categorical_num_classes=cols["K"], timesteps=T,
                          loss_lambda=0.8)
    g = torch.Generator().manual_seed(2023)
    data = torch.rand(max(4 * batch, 256), n_ch, L, generator=g)
    for c in cols["cat"]:
        data[:, c] = (data[:, c] > 0.5).float()
    et = ETDiff(diff, TensorSeriesDataset(data, cols["cat"]), train_batch_size=batch, gradient_accumulate_every=ga,
                diff_lr=8e-5, diff_num_steps=steps, ema_update_every=10, check_point_path="/tmp/bench_ckpt.pt",
                fuse_accumulation=fuse)
    for _ in range(warmup):
        et.train_step()
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(steps):
        loss = et.train_step()
    torch.cuda.synchronize()
    el = time.time() - t0
    assert torch.isfinite(loss).all()
    # roofline of the training half: the two kernels that carry the recurrent products of the layers above the first
    # (saving forward, BPTT sweep), each bracketed by HIP events on its launch stream in a few extra untimed steps
    from timediff_amd import _lib
    lib = _lib.lib()
    seqs = batch * ga if fuse else batch           # sequences per launch
    D = 2
    fwd_flops = seqs * D * 2.0 * L * 4 * H * (D * H + H)     # W_hh h + W_ih x, one layer, both directions
    bptt_flops = seqs * D * 2.0 * L * 4 * H * H               # W_hh^T d a (d x and d W are separate GEMM launches)
    kernels = {}
    graphed = bool(et._graphs) and all("graph" in e for e in et._graphs.values())
    et.graph_steps = False      # the probes below bracket kernels with events launch by launch; a replayed graph has none
    for kind, name, fl in ((_lib.PROFILE_LAYER_SAVE, "saving forward, layers above the first", fwd_flops),
                           (_lib.PROFILE_BPTT_SWEEP, "BPTT sweep, layers above the first", bptt_flops)):
        _lib.check(lib.td_profile_enable(kind))
        for _ in range(3):
            et.train_step()
        cnt, ms = C.c_int32(0), C.c_float(0)
        _lib.check(lib.td_profile_read(C.byref(cnt), C.byref(ms)))
        _lib.check(lib.td_profile_enable(0))
        avg_ms = ms.value / max(cnt.value, 1)
        tf = fl / (avg_ms * 1e-3) / 1e12 if avg_ms > 0 else 0.0
        kernels[name] = dict(family=_family_name(lib, et, seqs, L), launches_timed=cnt.value, avg_launch_ms=avg_ms,
                             flops_per_launch=fl, achieved_tflops=tf, frac_of_mfma_peak=tf / MFMA_BF16_PEAK_TFLOPS)
    width_ = len(cols["num"]) + sum(cols["K"])
    step_flops = 3.0 * denoiser_flops_per_seq(width_, H, N, D, L) * batch * ga
    roofline = dict(bound="mfma", peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", kernels=kernels,
                    whole_step_tflops=step_flops / (el / steps) / 1e12,
                    whole_step_frac=step_flops / (el / steps) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                    note="whole step = 3 x forward FLOPs x sequences / wall time per optimizer step")
    return dict(value=steps / el, unit="optimizer steps/s", ms_per_step=el / steps * 1e3, steps=steps, warmup=warmup, roofline=roofline,
                config=dict(workload=f"ETDiff eICU-shaped training step: global batch {batch}, grad-accum {ga}, L={L}, "
                                     f"3-layer bidirectional LSTM H={H}, T={T}, Adam + clip + EMA",
                            batch=batch, gradient_accumulate_every=ga,
                            accumulation="micro-batches concatenated into one launch" if fuse and ga > 1 else "loop",
                            launch="one captured graph per optimizer step" if graphed else "kernel by kernel"),
                sequences_per_s=steps * batch * ga / el, final_loss=float(loss))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3, help="cap on timed full chains (see --budget-s)")
    ap.add_argument("--warmup", type=int, default=1, help="cap on untimed full chains (at most one is run)")
    ap.add_argument("--budget-s", type=float, default=240.0, help="wall budget for warm-up + timed chains")
    ap.add_argument("--batch", type=int, default=16384, help="sequences per GPU per chain")
    ap.add_argument("--timesteps", type=int, default=1000)
    ap.add_argument("--hidden", type=int, default=256)
    ap.add_argument("--seq-len", type=int, default=272)
    ap.add_argument("--probe-steps", type=int, default=8, help="eager reverse steps timed with HIP events for the roofline")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-train-probe", action="store_true")
    ap.add_argument("--train-batch", type=int, default=32, help="training probe: batch per micro-batch (reference default 32)")
    ap.add_argument("--train-steps", type=int, default=100, help="timed optimizer steps of the small-batch training probes")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="bound on the CPU-oracle baseline sample")
    ap.add_argument("--backend", default="nccl", help="nccl (= RCCL, one rank per GPU; what the driver runs); gloo only "
                    "rehearses the multi-rank control flow with ranks sharing a GPU -- its numbers mean nothing")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    if args.backend != "nccl":       # rehearsal of the N > 1 control flow on a box with fewer GPUs than ranks
        local_rank %= max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(args.backend)

    from timediff_amd import _lib
    from timediff_amd.ops import DenoiserKernels, DiffusionKernels, SamplerKernels
    from timediff_amd.schedule import make_tables

    cols, L, T, H, N, D = EICU, args.seq_len, args.timesteps, args.hidden, 3, 2
    C_io = len(cols["num"]) + sum(cols["K"])
    sd = init_state(C_io, H, N, D, seed=2023)
    tables = make_tables(T, "cosine")
    den = DenoiserKernels({k: v.to(dev) for k, v in sd.items()})
    diff = DiffusionKernels(cols["num"], cols["cat"], cols["K"], L, T, tables, len(cols["num"]) + len(cols["cat"]), dev)
    sampler = SamplerKernels(den, diff)
    sampler.refresh()
    B = args.batch

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def chain(i):
        # every chain draws fresh patients: ids are disjoint across ranks and chains
        return sampler.sample(B, seed=2023, patient0=(i * world + rank) * B)

    def note(msg):
        """Progress on stderr (a chain takes ~30 s; the JSON line is the only thing on stdout)."""
        if rank == 0:
            print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)

    def agree_max(x):
        """The same number on every rank (max over ranks): all ranks take the same stop decision."""
        if world == 1:
            return x
        tt = torch.tensor([x], device=dev if args.backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        return float(tt.item())

    # untimed: one short pass through every kernel so allocations and code objects are in place
    pre = SamplerKernels(den, DiffusionKernels(cols["num"], cols["cat"], cols["K"], L, 4, make_tables(4, "cosine"),
                                               len(cols["num"]) + len(cols["cat"]), dev))
    pre.sample(B, seed=1)
    barrier()
    budget0 = time.time()
    est = None                       # seconds per chain, agreed over ranks
    warmup_done = 0
    if args.warmup > 0:              # at most ONE full untimed chain, whatever the cap says
        t0 = time.time()
        chain(-1)
        torch.cuda.synchronize()
        est = agree_max(time.time() - t0)
        warmup_done = 1
        note(f"warm-up chain of {B} sequences: {est:.1f} s")
    barrier()
    t0 = time.time()
    steps_done = 0
    while steps_done < args.steps:
        used = agree_max(time.time() - budget0)
        # the first timed chain always runs; later ones only if they fit the budget
        if steps_done > 0 and used + est > args.budget_s:
            break
        c0 = time.time()
        out = chain(steps_done)
        torch.cuda.synchronize()     # a chain is ~30 s: the sync between chains is noise, and it bounds the run
        est = agree_max(time.time() - c0)
        steps_done += 1
        note(f"timed chain {steps_done}: {est:.1f} s")
    barrier()
    elapsed = agree_max(time.time() - t0)
    assert torch.isfinite(out).all()
    value = world * B * steps_done / elapsed
    # the boundary's only host transfer, outside the timed region: the decoded (B, C, L) fp32 samples, D2H
    d0 = time.time()
    host = out.cpu()
    d2h_s = agree_max(time.time() - d0)
    del host

    # every rank reports what it saw: the first SCALE run proves that N ranks, N devices and N disjoint patient ranges took part
    mine = dict(rank=rank, local_rank=local_rank, world_size_env=world,
                world_size_backend=(dist.get_world_size() if world > 1 else 1), backend=(dist.get_backend() if world > 1 else None),
                device=torch.cuda.get_device_name(dev), device_index=torch.cuda.current_device(),
                pci_bus_id=getattr(torch.cuda.get_device_properties(dev), "pci_bus_id", None),
                first_timed_chain_patients=[rank * B, (rank + 1) * B - 1], chains_timed=steps_done)
    ranks_seen = [mine]
    if world > 1:
        ranks_seen = [None] * world
        dist.all_gather_object(ranks_seen, mine)

    result = None
    if rank == 0:
        # roofline probe of the dominant kernel (recurrent layers above the first: td_rnn_layer_kernel<8,32,0,0>):
        # eager reverse steps with a HIP event pair around each launch on its own stream
        L_ = _lib.lib()
        z, idx = diff.prior(B, seed=7)
        _lib.check(L_.td_profile_enable(_lib.PROFILE_LAYER_INFER))
        for s in range(args.probe_steps):
            t = T - 1 - s
            mo = den.forward(diff.build_input(z, idx), film=sampler._film[t:t + 1])
            diff.reverse_step_(mo, z, idx, t, seed=7)
        cnt, ms = C.c_int32(0), C.c_float(0)
        _lib.check(L_.td_profile_read(C.byref(cnt), C.byref(ms)))
        _lib.check(L_.td_profile_enable(0))
        avg_s = ms.value / max(cnt.value, 1) * 1e-3
        flops_launch = B * D * 2.0 * L * 4 * H * (D * H + H)       # one launch = one layer, both directions
        achieved = flops_launch / avg_s / 1e12
        # HBM traffic of that kernel per launch comes from separate rocprofv3 --pmc passes (FETCH_SIZE doubled as
        # MI355X_MICROARCH.md prescribes for gfx950, + WRITE_SIZE); bench.py cannot collect counters itself, so it
        # reports the committed measurement when it was taken at this batch size
        # (the probe's launches are the PLAIN variant of the upper-layer kernel: the tensor-level forward keeps the head as
        # its own kernel; the chain's top layer is the HEAD variant, its first layer the L0 variant: all three are in the file)
        traffic, variants = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "layer_kernel_pmc.json")) as f:
                pm = json.load(f)
            if pm.get("batch") == B and pm.get("seq_len") == L and pm.get("hidden") == H:
                traffic = pm["variants"]["plain"]["hbm_bytes_per_launch"]
                variants = {k: dict(hbm_bytes_per_launch=v["hbm_bytes_per_launch"], mfma_util=v["mfma_util"]) for k, v in pm["variants"].items()}
        except (OSError, ValueError, KeyError):
            pass
        roofline = dict(bound="mfma", kernel="td_rnn_layer_kernel<NU=8,KX=32,L0=0,GRU=0,SAVE=0,HEAD=0>", variant="plain",
                        achieved=achieved, peak=MFMA_BF16_PEAK_TFLOPS, unit="TFLOP/s", frac=achieved / MFMA_BF16_PEAK_TFLOPS,
                        traffic=traffic, traffic_unit="bytes per launch (PMC)", pmc_by_variant=variants, launches_timed=cnt.value,
                        avg_launch_ms=avg_s * 1e3, flops_per_launch=flops_launch)
        note(f"roofline probe: {cnt.value} launches, {avg_s * 1e3:.2f} ms each")
        base = None
        if world == 1 and not args.no_cpu_baseline:   # rank 0 at N = 1 only: the other ranks would wait on it
            tabs = {k: v.numpy() for k, v in tables.items()}
            base = cpu_baseline(sd, cols, L, T, tabs, args.cpu_seconds)
            base["parity"] = parity_gates(dev, den, diff, sampler, sd, cols, L, T, tabs, B)
        note("cpu baseline done" if base else "cpu baseline skipped")
        train = None
        if world == 1 and not args.no_train_probe:
            del out, z, idx, mo
            torch.cuda.empty_cache()
            train = {"reference_batch": train_probe(dev, cols, L, H, N, T, args.train_batch, 2, args.train_steps, 5),
                     "reference_batch_looped": train_probe(dev, cols, L, H, N, T, args.train_batch, 2, args.train_steps, 5, fuse=False),
                     "large_batch": train_probe(dev, cols, L, H, N, T, 2048, 1, 12, 4)}      # (warm-up > ETDiff.graph_warmup: the capture is not timed)
            # BASELINE.json configs[4]'s per-GPU shape (wide-hidden BRNN, 512 sequences per GPU and optimizer step): reported beside
            # the metric's own configurations, not part of `value`
            torch.cuda.empty_cache()
            train["wide_hidden"] = train_probe(dev, cols, L, 512, N, T, 512, 1, 10, 4)
        fl = denoiser_flops_per_seq(C_io, H, N, D, L) * T
        result = {
            "metric": "synthetic patient sequences/sec (full T-step reverse chain)", "value": value,
            "unit": "sequences/s", "n_gpus": world, "steps": steps_done, "warmup": warmup_done,
            "steps_requested": args.steps, "warmup_requested": args.warmup, "budget_s": args.budget_s,
            "ms_per_step": elapsed / steps_done * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "ETDiff eICU-shaped sampler: C_io=14 (4 continuous + 5 binary), L=272, "
                                   f"3-layer bidirectional LSTM H={H}, T={T}, random-init weights, Philox noise",
                       "batch_per_gpu": B, "sequences_per_step": world * B, "parallelism": f"patient-sharded x{world}",
                       "graph": True},
            "d2h_s_per_chain": d2h_s, "value_including_d2h": world * B * steps_done / (elapsed + steps_done * d2h_s),
            "denoiser_tflops_whole_chain": value * fl / 1e12,
            "roofline": roofline, "cpu_baseline": base, "train": train, "ranks_seen": ranks_seen,
        }
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result), flush=True)


if __name__ == "__main__":
    main()

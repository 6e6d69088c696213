"""ETDiff trainer with the reference's interface (models/ETDiff/et_diff.py:21-264): same constructor
keywords, `.train()`, `.save_weights()`, `.load(path)`, `.ema.ema_model`, `.sample_min/.sample_max`,
`.step`, and the same checkpoint dict {"diff_step", "diff_model", "opt", "ema", "scaler"}.

What differs underneath (MI355X-first, same arithmetic):
  * no accelerate / DDP wrappers: one process per GPU; with WORLD_SIZE > 1 the global batch is split
    across ranks (split_batches=True semantics, et_diff.py:45) and the flat fp32 gradient buffer is
    all-reduced (mean) over RCCL once per optimizer step on a side stream, one slice per recurrent layer,
    each started as soon as that layer's gradient kernels are enqueued (overlapped with the BPTT below it);
  * all parameters, gradients, Adam moments and EMA weights live in flat fp32 buffers: grad-norm, clip,
    Adam and the EMA update are two fused kernel launches (td_grad_norm, td_clip_adam_ema);
  * the loss is read back once per optimizer step, not once per micro-batch.
EMA: the reference uses the third-party `ema_pytorch.EMA` (neither vendored nor version-pinned, so parity
is unpinned); this class follows that package's documented defaults: no averaging until step 100
(weights are copied), then decay = min(beta, 1 - (1 + (step - 101))^(-2/3)), applied every
`update_every` optimizer steps; the "ema" sub-dict keeps the `ema_model.` key prefix.
"""
import copy
import ctypes as C
import logging
import os
import sys
import timeit

import numpy as np
import torch
import torch.nn as nn
from torch.utils.data import DataLoader

from . import _lib
from ._lib import check, lib, ptr
from .data import exists
from .parallel import layer_slices, uncovered_ranges

logging.basicConfig(filename=None, format="[%(levelname)s:%(asctime)s] %(message)s", level=logging.INFO)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _batch_passthrough(batch):
    """collate_fn for datasets whose __getitems__ already returns the (B, C, L) batch tensor."""
    return batch


def _flatten_params_(module):
    """Re-points every parameter of `module` into one contiguous fp32 buffer; returns (flat, params)."""
    params = [p for p in module.parameters()]
    flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=params[0].device)
    off = 0
    for p in params:
        n = p.numel()
        flat[off:off + n].copy_(p.data.reshape(-1))
        p.data = flat[off:off + n].view_as(p)
        off += n
    return flat, params


def _cycle_epochs(dl, sampler):
    """cycle() (utils.py:107-110) that also advances a DistributedSampler's epoch each time the loader wraps, so
    every pass over the data uses a new permutation (the same one on all ranks)."""
    epoch = 0
    while True:
        if sampler is not None:
            sampler.set_epoch(epoch)
        for data in dl:
            yield data
        epoch += 1


class EMA(nn.Module):
    """Exponential moving average of the diffusion module (interface of ema_pytorch.EMA as the reference
    uses it: EMA(model, beta=, update_every=), .ema_model, .update(), state_dict with `ema_model.*`)."""

    def __init__(self, model, beta=0.995, update_every=10, update_after_step=100, inv_gamma=1.0, power=2 / 3):
        super().__init__()
        self.beta, self.update_every, self.update_after_step = beta, update_every, update_after_step
        self.inv_gamma, self.power = inv_gamma, power
        self.ema_model = copy.deepcopy(model)
        self.ema_model.requires_grad_(False)
        self.register_buffer("initted", torch.tensor(False))
        self.register_buffer("step", torch.tensor(0))

    def current_decay(self, step):
        epoch = max(step - self.update_after_step - 1, 0)
        if epoch <= 0:
            return 0.0
        return min(max(1 - (1 + epoch / self.inv_gamma) ** -self.power, 0.0), self.beta)

    def _host_counters(self):
        """The step counter and the initted flag live in buffers (they are part of the checkpoint's `ema` dict); reading
        them back every optimizer step would drain the GPU queue, so the host keeps its own copy and only writes the
        buffers.  A load_state_dict invalidates the copy."""
        if getattr(self, "_host", None) is None:
            self._host = [int(self.step.item()), bool(self.initted.item())]
        return self._host

    def _load_from_state_dict(self, *args, **kwargs):
        self._host = None
        return super()._load_from_state_dict(*args, **kwargs)

    def plan_update(self):
        """Advances the step counter and says what this step does to the EMA weights:
        (mode, decay) with mode 0 = nothing, 1 = copy the online weights, 2 = lerp."""
        host = self._host_counters()
        host[0] += 1
        step = host[0]
        self.step.fill_(step)
        if step % self.update_every != 0:
            return 0, 0.0
        if step <= self.update_after_step or not host[1]:
            host[1] = True
            self.initted.fill_(True)
            return 1, 0.0
        return 2, self.current_decay(step)


class ETDiff(nn.Module):
    def __init__(self, diffusion_model, dataset, *, train_batch_size=16, gradient_accumulate_every=1, diff_lr=1e-4,
                 diff_num_steps=100000, ema_update_every=10, ema_decay=0.995, adam_betas=(0.9, 0.99),
                 sample_every=1000, amp=False, mixed_precision_type="fp16", split_batches=True, wandb=None,
                 check_point_path="results/checkpoint.pt", run_id=2023, fuse_accumulation=True):
        super().__init__()
        if amp:
            raise NotImplementedError("amp is hard-wired to False by the reference's CLI (etdiff_train.py:270)")
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if not torch.cuda.is_available():
            raise RuntimeError("ETDiff trains on the MI355X only (no CPU fallback)")
        torch.cuda.set_device(local_rank)
        self._device = torch.device("cuda", local_rank)
        # TIMEDIFF_DP_FORCE=1 runs the data-parallel code path (process group, communication stream, per-layer
        # all-reduce behind events) even with ONE rank: RCCL accepts a single-rank communicator, which lets a
        # one-GPU box execute the production (nccl) branch (tests/test_zz_gpu_data_parallel.py)
        self._dp = self.world > 1 or os.environ.get("TIMEDIFF_DP_FORCE", "0") == "1"
        if self._dp and not torch.distributed.is_initialized():
            # "nccl" is RCCL on ROCm; TIMEDIFF_DIST_BACKEND=gloo lets several ranks share one GPU (tests)
            backend = os.environ.get("TIMEDIFF_DIST_BACKEND", "nccl")
            torch.distributed.init_process_group(backend, rank=self.rank, world_size=self.world,
                                                 **({"device_id": self._device} if backend == "nccl" else {}))
        print(f"*** Using device: {self._device} ***", file=sys.stderr)

        self.timer = timeit.default_timer
        self.wandb = wandb
        self.check_point_path = check_point_path
        self.sample_min = dataset.min.numpy()
        self.sample_max = dataset.max.numpy()

        self.model = diffusion_model.to(self._device)
        self.channels = diffusion_model.channels
        self.sample_every = sample_every
        self.batch_size = train_batch_size
        self.gradient_accumulate_every = gradient_accumulate_every
        # The reference sums the gradients of gradient_accumulate_every micro-batch losses, each divided by that count
        # (et_diff.py:231-237): with micro-batches of equal size that is the gradient of the mean loss over their
        # concatenation, so they run as ONE launch (the recurrent kernels at the reference's batch of 32 are bound by
        # step latency, not throughput: two batch blocks take the time of one).  False restores the loop.
        self.fuse_accumulation = bool(fuse_accumulation)
        self.diff_num_steps = diff_num_steps
        self.run_id = run_id

        # split_batches=True: train_batch_size is the global batch, every rank takes its slice
        assert train_batch_size % self.world == 0, "global batch must divide evenly across ranks"
        self.local_batch = train_batch_size // self.world
        sampler = None
        if self.world > 1:
            # every rank must permute with the SAME seed (else the shards overlap): run_id is int(time.time()) per
            # process in the CLI, so rank 0's value is broadcast
            sampler = torch.utils.data.distributed.DistributedSampler(dataset, self.world, self.rank, shuffle=True,
                                                                      seed=self._shared_int(run_id))
        # no loader workers: the dataset is one in-memory tensor (a batch is one gather: TimeSeriesDataset.__getitems__), and a worker would be forked from a process that has
        # already initialised HIP (a forked child can inherit a runtime lock that is held and never released: a rare hang)
        dl = DataLoader(dataset, batch_size=self.local_batch, shuffle=sampler is None, sampler=sampler, pin_memory=True,
                        num_workers=0, drop_last=self.world > 1,
                        collate_fn=_batch_passthrough if hasattr(dataset, "__getitems__") else None)
        self.dl = _cycle_epochs(dl, sampler)

        # flat buffers: weights, gradients, Adam moments, EMA weights
        self.lr, self.betas, self.eps = diff_lr, adam_betas, 1e-8
        self.flat_param, self.params = _flatten_params_(self.model)
        self.flat_grad = torch.zeros_like(self.flat_param)
        self.exp_avg = torch.zeros_like(self.flat_param)
        self.exp_avg_sq = torch.zeros_like(self.flat_param)
        off = 0
        for p in self.params:
            p.grad = self.flat_grad[off:off + p.numel()].view_as(p)
            off += p.numel()
        self.ema = EMA(self.model, beta=ema_decay, update_every=ema_update_every)
        self.ema.to(self._device)
        self.flat_ema, _ = _flatten_params_(self.ema.ema_model)
        self._norm = torch.zeros(1, dtype=torch.float32, device=self._device)
        self._norm_ws = torch.empty(lib().td_grad_norm_workspace_bytes(), dtype=torch.uint8, device=self._device)
        self._comm_stream = torch.cuda.Stream(device=self._device) if self._dp else None
        # The denoiser's backward accumulates straight into the flat gradient buffer (no AccumulateGrad pass); with
        # several ranks each recurrent layer's slice is all-reduced as soon as its gradient kernels are enqueued,
        # behind an event, on the communication stream, while BPTT goes on in the layers below.
        names = [n for n, _ in self.model.named_parameters()]
        den = getattr(self.model, "model", None)
        self._sink = None
        self._reduced = []
        self._overlap = os.environ.get("TIMEDIFF_DP_OVERLAP", "1") != "0"
        if den is not None and hasattr(den, "kernels"):
            by_name = dict(self.model.named_parameters())
            self._sink = {"tensors": {n: by_name["model." + n].grad for n, _ in den.named_parameters()}, "on_layer_done": None}
            den._grad_sink = self._sink
            self._n_layers = den.rnn.rnn.num_layers
            self._layer_slices = layer_slices(names, [p.numel() for p in self.params], self._n_layers)
        self.opt_step = 0
        self.step = 0
        # the batches the trainer hands to the diffusion module are its own device copies: the reference's in-place
        # normalisation of the caller's batch (mixed_diffusion.py:189-194) would be four kernels that nobody observes
        self.model._batch_is_private = True
        # One process, one fused batch, no dropout: the whole optimizer step (q_sample ... Adam/EMA, about sixty small
        # launches) is captured once per (batch, L) signature and replayed as ONE graph launch; what changes from step to
        # step (the noise seed, Adam's bias corrections, what the EMA does) is read by the kernels from a device record
        # (td_step_record) that is rewritten before each replay.  TIMEDIFF_TRAIN_GRAPH=0 keeps the launch-by-launch path;
        # TIMEDIFF_TRAIN_GRAPH_WARMUP = optimizer steps run launch by launch before a signature is captured.
        self.graph_steps = os.environ.get("TIMEDIFF_TRAIN_GRAPH", "1") != "0"
        self.graph_warmup = int(os.environ.get("TIMEDIFF_TRAIN_GRAPH_WARMUP", "2"))
        self._graphs = {}
        self._rec = torch.zeros(C.sizeof(_lib.StepRecord), dtype=torch.uint8, device=self._device)

    def _shared_int(self, value):
        """rank 0's `value` on every rank."""
        import torch.distributed as dist
        if self.world == 1 or not dist.is_initialized():
            return int(value)
        box = torch.tensor([int(value)], dtype=torch.int64, device=self._device if dist.get_backend() == "nccl" else "cpu")
        dist.broadcast(box, src=0)
        return int(box.item())

    @property
    def device(self):
        return self._device

    @property
    def is_main_process(self):
        return self.rank == 0

    # -- checkpoint (et_diff.py:91-105, :208-216) --------------------------------------------------
    def _opt_state_dict(self):
        """torch.optim.Adam.state_dict() layout, built from the flat moments."""
        state, off = {}, 0
        for i, p in enumerate(self.params):
            n = p.numel()
            state[i] = {"step": torch.tensor(float(self.opt_step)),
                        "exp_avg": self.exp_avg[off:off + n].view_as(p).clone(),
                        "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p).clone()}
            off += n
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "params": list(range(len(self.params)))}
        return {"state": state if self.opt_step > 0 else {}, "param_groups": [group]}

    def _load_opt_state_dict(self, sd):
        off = 0
        for i, p in enumerate(self.params):
            n = p.numel()
            if i in sd["state"]:
                st = sd["state"][i]
                self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                self.opt_step = int(float(st["step"]))
            off += n
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]

    def save_weights(self, block=True):
        """The reference's checkpoint dict (et_diff.py:208-216), written without stalling the training loop: the device
        state is snapshotted in stream order (four device-to-device copies of the flat weight / EMA / Adam buffers plus
        the few registered buffers), copied to pinned host memory on a side stream, and a writer thread builds the
        dict, serialises it to a temporary file and renames it into place.  block=True (the default, what the
        reference's synchronous torch.save gives a caller) returns once the file is there; the training loop passes
        block=False and joins the writer before the next checkpoint and at the end of training."""
        import threading
        self.wait_for_checkpoint()
        dev = torch.cuda.current_stream()
        flat = {"param": self.flat_param, "ema": self.flat_ema, "m": self.exp_avg, "v": self.exp_avg_sq}
        snap = {k: v.detach().clone() for k, v in flat.items()}

        def layout(module, flat_t):
            """state_dict entries in order: ("flat", offset, shape) for views into the flat buffer, ("own", clone) otherwise"""
            lo, hi = flat_t.data_ptr(), flat_t.data_ptr() + flat_t.numel() * 4
            out = []
            for name, t in module.state_dict().items():
                if t.is_cuda and t.dtype == torch.float32 and lo <= t.data_ptr() < hi:
                    out.append((name, "flat", (t.data_ptr() - lo) // 4, tuple(t.shape)))
                else:
                    out.append((name, "own", t.detach().clone(), None))
            return out
        plan_model, plan_ema = layout(self.model, self.flat_param), layout(self.ema, self.flat_ema)
        ready = torch.cuda.Event()
        ready.record(dev)
        if getattr(self, "_ckpt_stream", None) is None:
            self._ckpt_stream = torch.cuda.Stream(device=self._device)
            self._ckpt_host = {k: torch.empty(v.shape, dtype=v.dtype, pin_memory=True) for k, v in snap.items()}
            self._ckpt_small = {}       # pinned copies of the registered buffers, by (which dict, name)
        with torch.cuda.stream(self._ckpt_stream):
            self._ckpt_stream.wait_event(ready)
            for k, v in snap.items():
                self._ckpt_host[k].copy_(v, non_blocking=True)
                v.record_stream(self._ckpt_stream)
            for which, plan in enumerate((plan_model, plan_ema)):
                for i, (name, kind, a, b) in enumerate(plan):
                    if kind == "own":
                        h = a
                        if a.is_cuda:
                            h = self._ckpt_small.get((which, name))
                            if h is None or h.shape != a.shape or h.dtype != a.dtype:
                                h = self._ckpt_small[(which, name)] = torch.empty(a.shape, dtype=a.dtype, pin_memory=True)
                            h.copy_(a, non_blocking=True)
                            a.record_stream(self._ckpt_stream)
                        plan[i] = (name, "host", h, None)
            done = torch.cuda.Event()
            done.record(self._ckpt_stream)
        meta = dict(step=self.step, opt_step=self.opt_step, lr=self.lr, betas=tuple(self.betas), eps=self.eps,
                    shapes=[tuple(p.shape) for p in self.params], path=self.check_point_path)
        host = self._ckpt_host

        def build(plan, flat_h):
            from collections import OrderedDict
            return OrderedDict((name, flat_h[a:a + int(np.prod(b, dtype=np.int64))].view(b).clone() if kind == "flat" else a.clone())
                               for name, kind, a, b in plan)

        def write():
            done.synchronize()
            state, off = {}, 0
            for i, shape in enumerate(meta["shapes"]):
                n = int(np.prod(shape, dtype=np.int64))
                state[i] = {"step": torch.tensor(float(meta["opt_step"])), "exp_avg": host["m"][off:off + n].view(shape).clone(),
                            "exp_avg_sq": host["v"][off:off + n].view(shape).clone()}
                off += n
            group = {"lr": meta["lr"], "betas": meta["betas"], "eps": meta["eps"], "weight_decay": 0, "amsgrad": False,
                     "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                     "params": list(range(len(meta["shapes"])))}
            weights = {"diff_step": meta["step"], "diff_model": build(plan_model, host["param"]),
                       "opt": {"state": state if meta["opt_step"] > 0 else {}, "param_groups": [group]},
                       "ema": build(plan_ema, host["ema"]), "scaler": None}
            tmp = meta["path"] + ".tmp"
            torch.save(weights, tmp)
            os.replace(tmp, meta["path"])

        self._ckpt_error = None

        def guarded():
            try:
                write()
            except BaseException as e:      # surfaces in wait_for_checkpoint
                self._ckpt_error = e
        self._ckpt_thread = threading.Thread(target=guarded, name="timediff-checkpoint", daemon=False)
        self._ckpt_thread.start()
        if block:
            self.wait_for_checkpoint()

    def wait_for_checkpoint(self):
        """Joins the checkpoint writer, if one is in flight, and re-raises what it raised."""
        th = getattr(self, "_ckpt_thread", None)
        if th is not None:
            th.join()
            self._ckpt_thread = None
            if self._ckpt_error is not None:
                err, self._ckpt_error = self._ckpt_error, None
                raise err

    def load(self, path):
        # the checkpoint dict holds tensors, dicts, lists, tuples, numbers and None only (et_diff.py:208-216): the
        # restricted unpickler loads it, and nothing in a user-supplied --eval_path gets executed
        data = torch.load(path, map_location=self.device, weights_only=True)
        self.model.load_state_dict(data["diff_model"])     # copies into the flat buffer in place
        self.step = data["diff_step"]
        self._load_opt_state_dict(data["opt"])
        self.ema.load_state_dict(data["ema"], strict=False)

    # -- one optimizer step --------------------------------------------------------------------------
    def _reduce_slice(self, a, b):
        """Mean over ranks of flat_grad[a:b], enqueued on the communication stream."""
        import torch.distributed as dist
        with torch.cuda.stream(self._comm_stream):
            g = self.flat_grad[a:b]
            if dist.get_backend() == "gloo":
                # test rigs only (several ranks on one GPU): stage through the host on this stream instead of going
                # through gloo's own CUDA streams and events
                h = g.cpu()
                dist.all_reduce(h, op=dist.ReduceOp.SUM)
                g.copy_(h.div_(self.world))
            else:
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
                g.div_(self.world)

    def _layer_reduce(self, layer):
        """td_rnn_backward_cb's callback on the last micro-batch: layer `layer`'s gradient kernels are all enqueued."""
        if layer >= self._n_layers:
            return      # LayerNorm / fc: reduced with the rest of the small tensors at the end
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        self._comm_stream.wait_event(ev)
        a, b = self._layer_slices[layer]
        self._reduce_slice(a, b)
        self._reduced.append((a, b))

    def _all_reduce_grads(self):
        """Whatever the per-layer reductions have not covered (everything when overlap is off), then join."""
        if not self._dp:
            return
        self._comm_stream.wait_stream(torch.cuda.current_stream())
        for a, b in uncovered_ranges(self.flat_grad.numel(), self._reduced):
            self._reduce_slice(a, b)
        self._reduced = []
        torch.cuda.current_stream().wait_stream(self._comm_stream)

    def _optimizer_step(self, update_ema):
        n = self.flat_param.numel()
        check(lib().td_grad_norm(ptr(self.flat_grad), n, ptr(self._norm), ptr(self._norm_ws), _stream()))
        self.opt_step += 1
        mode, decay = self.ema.plan_update() if update_ema else (0, 0.0)
        check(lib().td_clip_adam_ema(ptr(self.flat_param), ptr(self.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                     ptr(self.flat_ema), n, ptr(self._norm), 1.0, self.lr, self.betas[0], self.betas[1],
                                     self.eps, self.opt_step, mode, decay, _stream()))
        # in-place kernel writes do not bump torch's version counters: tell the denoisers their caches are stale
        self._bump_versions()
        self.flat_grad.zero_()

    def _bump_versions(self):
        for mod in (self.model, self.ema.ema_model):
            den = getattr(mod, "model", None)
            if den is not None and hasattr(den, "_kernels_version"):
                den._kernels_version = None
            if hasattr(mod, "_film_version"):
                mod._film_version = None

    # -- the optimizer step as one graph launch ------------------------------------------------------
    def _graph_eligible(self):
        """A captured step holds no host decision: one rank (no all-reduce callbacks), no inter-layer dropout (its seed is a
        host draw), the denoiser on this library's kernels.  Two recurrent layers at least: a captured cluster-kernel launch
        carries its hand-off epoch baked in, and stale tags are told apart only because the launch before it on the same
        exchange buffer had another epoch (csrc/sampler.hip says the same of the sampler's graph)."""
        den = getattr(self.model, "model", None)
        return (self.graph_steps and not self._dp and self._sink is not None and self._n_layers >= 2
                and not (den.training and den.dropout > 0))

    def _capture_step(self, batches):
        den = self.model.model
        xs = [torch.empty(b.shape, dtype=b.dtype, device=self.device) for b in batches]
        for x, b in zip(xs, batches):
            x.copy_(b, non_blocking=True)
        # what the kernels keep between calls is allocated here, outside the graph's memory pool, and the entry holds on
        # to it: a later, larger eager batch may replace the denoiser's workspace, this graph keeps using its own
        k = den.kernels()
        keep = (k, k.packed) + tuple(k.reserve_train(max(b.shape[0] for b in batches), batches[0].shape[2]))
        self.model.kernels()
        den._kernels_version = None       # the weight repack is part of every replay
        n, ga = self.flat_param.numel(), len(batches)
        graph = torch.cuda.CUDAGraph()
        self.model._seed_dev = self._rec
        try:
            with torch.cuda.graph(graph):
                total = None
                for i, x in enumerate(xs):
                    self.model._seed_base = i       # micro-batch i draws with seed = record's seed + i
                    loss = self.model(x)
                    if ga > 1:
                        loss = loss / ga
                    total = loss.detach() if total is None else total + loss.detach()
                    loss.backward()
                check(lib().td_grad_norm(ptr(self.flat_grad), n, ptr(self._norm), ptr(self._norm_ws), _stream()))
                check(lib().td_clip_adam_ema_rec(ptr(self.flat_param), ptr(self.flat_grad), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                                 ptr(self.flat_ema), n, ptr(self._norm), 1.0, self.betas[0], self.betas[1], self.eps,
                                                 ptr(self._rec), _stream()))
                self.flat_grad.zero_()
        finally:
            self.model._seed_dev = None
            self.model._seed_base = 0
        return {"graph": graph, "x": xs, "loss": total, "keep": keep}

    def _train_step_graphed(self, batches):
        """-> the step's loss, or None while this signature (the micro-batch shapes) is still in its launch-by-launch
        warm-up."""
        # (Adam's betas and eps are kernel arguments of the captured step: a checkpoint that changes them gets a new graph)
        key = tuple((tuple(b.shape), b.dtype) for b in batches) + (tuple(self.betas), self.eps)
        ent = self._graphs.get(key)
        if ent is None:
            ent = self._graphs[key] = {"seen": 0}
        if "graph" not in ent and ent["seen"] < self.graph_warmup:
            ent["seen"] += 1
            return None
        self.opt_step += 1
        mode, decay = self.ema.plan_update()
        seed = int(torch.randint(0, 2 ** 62, (1,)).item())      # torch's CPU generator: torch.manual_seed repeats a run
        check(lib().td_step_record_set(ptr(self._rec), seed, self.lr, self.betas[0], self.betas[1], self.opt_step, mode, decay,
                                       _stream()))
        if "graph" not in ent:
            ent.update(self._capture_step(batches))     # capturing runs nothing: the first replay below is this step
        else:
            for x, b in zip(ent["x"], batches):
                x.copy_(b, non_blocking=True)
        ent["graph"].replay()
        self._bump_versions()
        self.step += 1
        return ent["loss"].clone()

    def train_step(self):
        """gradient_accumulate_every micro-batches + one optimizer step; -> the step's loss (device scalar)."""
        ga = self.gradient_accumulate_every
        batches = [next(self.dl) for _ in range(ga)]
        if self.fuse_accumulation and ga > 1 and all(b.shape[0] == batches[0].shape[0] for b in batches):
            batches, ga = [torch.cat(batches)], 1       # mean over the concatenation = sum of the micro-batch means / ga
        if self._graph_eligible():
            loss = self._train_step_graphed(batches)
            if loss is not None:
                return loss
        total = None
        for i, data in enumerate(batches):
            if self._sink is not None:
                overlap = self._dp and self._overlap and i == ga - 1
                self._sink["on_layer_done"] = self._layer_reduce if overlap else None
            data = data.to(self.device, non_blocking=True)
            loss = self.model(data)
            if ga > 1:
                loss = loss / ga
            total = loss.detach() if total is None else total + loss.detach()
            loss.backward()
        self._all_reduce_grads()
        self._optimizer_step(update_ema=True)
        self.step += 1
        return total

    def train(self):
        from tqdm import tqdm
        last_log = self.timer()
        logging.info("********** START TRAINING DIFFUSION MODEL **********")
        with tqdm(initial=self.step, total=self.diff_num_steps, disable=not self.is_main_process) as pbar:
            pending = None      # the loss is read one step late: the next step is enqueued before the host waits
            while self.step < self.diff_num_steps:
                loss = self.train_step()
                total_loss = (pending if pending is not None else loss).item()
                _lib.check_async()      # the read-back waited for that step's kernels
                pending = loss
                pbar.set_description(f"loss: {total_loss:.4f}")
                if exists(self.wandb) and self.timer() - last_log > 60:
                    self.wandb.log({"total_loss": total_loss})
                    last_log = self.timer()
                if self.is_main_process and self.step != 0 and self.step % self.sample_every == 0:
                    self.save_weights(block=False)      # the writer runs beside the next steps
                pbar.update(1)
        if self.is_main_process:
            self.save_weights()
        print("========================= TRAINING COMPLETE =========================")

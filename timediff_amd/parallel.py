"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
MI355X node, "gloo" in the CPU tests).

* Sampling shards by patient: chains never interact (mixed_diffusion.py:437-446), so rank r draws the
  patients [start, stop) of the global range with Philox streams keyed by the GLOBAL patient id -- results
  do not depend on the number of ranks and there is no data-path collective.
* Training is data parallel: the global batch is split across ranks (split_batches=True, et_diff.py:45)
  and the flat fp32 gradient buffer is averaged with one all-reduce per optimizer step.
"""
import torch
import torch.distributed as dist


def patient_shard(total, rank, world):
    """[start, stop) of `total` patients owned by `rank`: contiguous, disjoint, covering, sizes differ by <= 1."""
    base, rem = divmod(total, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def all_reduce_mean_(flat, group=None):
    """In-place mean of a flat gradient buffer over the ranks of `group`."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return flat
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(dist.get_world_size(group))
    return flat


def gather_samples(local, total, group=None):
    """Rank 0 receives the (total, C, L) concatenation of every rank's shard (host tensors), others None."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    sizes = [patient_shard(total, r, world) for r in range(world)]
    if rank == 0:
        parts = [local] + [torch.empty((b - a,) + tuple(local.shape[1:]), dtype=local.dtype) for a, b in sizes[1:]]
        for r in range(1, world):
            dist.recv(parts[r], src=r, group=group)
        return torch.cat(parts, dim=0)
    dist.send(local.contiguous(), dst=0, group=group)
    return None


def layer_slices(names, sizes, layers, prefix="model.rnn.rnn."):
    """[(begin, end)] per recurrent layer in a flat buffer that holds the parameters `names` (sizes `sizes`) back to
    back: nn.LSTM / nn.GRU register weight_ih, weight_hh, bias_ih, bias_hh of a layer (then its _reverse twins)
    contiguously, so each layer is ONE slice = one all-reduce."""
    off, spans = 0, [None] * layers
    for name, n in zip(names, sizes):
        if name.startswith(prefix) and "_l" in name:
            l = int(name[len(prefix):].split("_l")[1].split("_")[0])
            a, b, tot = spans[l] or (off, off, 0)
            spans[l] = (min(a, off), max(b, off + n), tot + n)
        off += n
    out = []
    for l, sp in enumerate(spans):
        assert sp is not None and sp[1] - sp[0] == sp[2], f"layer {l}'s parameters are not contiguous in the flat buffer"
        out.append((sp[0], sp[1]))
    return out


def uncovered_ranges(n, done):
    """The parts of [0, n) that no (begin, end) of `done` covers, in order."""
    out, pos = [], 0
    for a, b in sorted(done):
        if a > pos:
            out.append((pos, a))
        pos = max(pos, b)
    if pos < n:
        out.append((pos, n))
    return out

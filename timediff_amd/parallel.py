"""Multi-GPU plumbing: one process per GPU, torch.distributed (backend "nccl" = RCCL over xGMI on the
MI355X node, "gloo" in the CPU tests).

* Sampling shards by patient: chains never interact (mixed_diffusion.py:437-446), so rank r draws the
  patients [start, stop) of the global range with Philox streams keyed by the GLOBAL patient id -- results
  do not depend on the number of ranks and there is no data-path collective; every rank writes its rows of the
  output `.npy` itself (SampleWriter).
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


class SampleWriter:
    """Streaming, multi-rank writer of the `.npy` the reference's collect_samples + np.save produce
    (etdiff_train.py:150-193; read back by eval_samples.py:219-231 with np.load).

    The reference concatenates every chain on the host and saves once; at 1 M eICU-shaped patients that is a
    9.8 GB host tensor funnelled through one process.  Here rank 0 creates the file with its final header and
    size (np.lib.format.open_memmap), every rank maps it and writes the rows [start, stop) of the chains it drew,
    already de-normalised -- no inter-rank traffic, no host concatenation, and the D2H copy + write of one chunk
    overlaps the next chunk's chain on the GPU (`put` is called with a pinned host tensor while the next chain
    runs).  The array shape is the reference's `torch.cat(...).squeeze()` shape: size-1 axes are dropped.
    """

    def __init__(self, path, total, channels, seq_len, min=None, max=None, rank=0, world=1, barrier=None):
        import numpy as np
        self.path, self.total, self.rank, self.world = path, int(total), rank, world
        self._barrier = barrier or (dist.barrier if (dist.is_available() and dist.is_initialized() and world > 1) else (lambda: None))
        self.full_shape = (self.total, int(channels), int(seq_len))
        squeezed = tuple(n for n in self.full_shape if n != 1)
        self._scale = None
        if min is not None and max is not None:
            mn = np.asarray(min, dtype=np.float32).reshape(1, -1, 1)
            mx = np.asarray(max, dtype=np.float32).reshape(1, -1, 1)
            self._scale = (mx - mn, mn)       # helpers/utils.py:182: x * (max - min) + min
        if rank == 0:
            import os
            os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
            mm = np.lib.format.open_memmap(path, mode="w+", dtype=np.float32, shape=squeezed)
            mm.flush()
            del mm
        self._barrier()                     # the file exists with its final size before anyone maps it
        self._mm = np.load(path, mmap_mode="r+")
        assert self._mm.shape == squeezed and self._mm.dtype == np.float32
        self._view = self._mm.reshape(self.full_shape)      # a view: size-1 axes restored
        self.rows_written = 0

    def put(self, start, chunk):
        """Rows [start, start + len(chunk)) <- chunk ((n, C, L) fp32 host tensor or ndarray, normalised samples)."""
        import numpy as np
        a = chunk.numpy() if isinstance(chunk, torch.Tensor) else np.asarray(chunk)
        n = a.shape[0]
        assert 0 <= start and start + n <= self.total and a.shape[1:] == self.full_shape[1:], (start, a.shape, self.full_shape)
        dst = self._view[start:start + n]
        if self._scale is None:
            dst[...] = a
        else:
            np.multiply(a, self._scale[0], out=dst)
            dst += self._scale[1]
        self.rows_written += n

    def close(self):
        self._mm.flush()
        self._view = None
        self._mm = None
        self._barrier()                     # every rank's rows are in the file
        return self.path


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

"""Host-side data helpers (dataset caching, min-max normalisation) with the names and behaviour of the
reference's models/ETDiff/utils.py.

Data handling here is glue around the hot path (dataset caching, min-max normalisation); the
schedules live in timediff_amd/schedule.py.
"""
import os

import numpy as np
import torch
from torch.utils.data import Dataset

from .schedule import cosine_beta_schedule, linear_beta_schedule  # noqa: F401  (re-exported, utils.py:75-87)


def _load_mapped(path):
    """A tensor file opened memory-mapped where the file format allows it (zip-format torch.save): nothing is read until it
    is touched, so a 10 GB cache costs no start-up time and no resident copy.  weights_only: a dataset file is data."""
    try:
        return torch.load(path, mmap=True, weights_only=True)
    except (RuntimeError, ValueError):      # legacy (non-zip) files cannot be mapped
        return torch.load(path, weights_only=True)


class _ChannelStats:
    """Per-channel running sum / count / min / max over (samples, time) of the non-NaN entries, in float64."""

    def __init__(self, channels):
        self.total = np.zeros(channels, np.float64)
        self.count = np.zeros(channels, np.int64)
        self.low = np.full(channels, np.inf)
        self.high = np.full(channels, -np.inf)

    def add(self, chunk):
        x = chunk.double().numpy()
        ok = ~np.isnan(x)
        self.total += np.where(ok, x, 0.0).sum(axis=(0, 2))
        self.count += ok.sum(axis=(0, 2))
        self.low = np.minimum(self.low, np.where(ok, x, np.inf).min(axis=(0, 2)))
        self.high = np.maximum(self.high, np.where(ok, x, -np.inf).max(axis=(0, 2)))

    @property
    def mean(self):
        return self.total / np.maximum(self.count, 1)


class TimeSeriesDataset(Dataset):
    """The cached dataset of the reference (utils.py:10-51): `<root>/<data_name>/{data,min,max}.pt`, an (N, C, L) float
    tensor whose non-indicator channels are min-max normalised per channel, truncated to `cut_length` steps.

    Built for datasets the size of what the sampler writes (1 M patients = 9.8 GB): the cache is opened memory-mapped,
    a batch is ONE gather of its rows (`__getitems__`, which torch's DataLoader uses when present), and building the cache
    from a raw tensor streams it twice in chunks of samples -- pass 1: per-channel mean / min / max of the non-NaN
    entries; pass 2: NaN -> channel mean, then (x - min) / (max - min) on the non-indicator channels -- into a
    memory-mapped staging file, so neither pass holds more than a chunk.  A NaN takes the channel mean, which lies between
    the channel's min and max, so the statistics of pass 1 are those of the filled data (the reference fills first and
    then takes min / max)."""

    CHUNK = 16384      # samples per streamed chunk

    def __init__(self, data_name, categorical_cols, cut_length, load_path=None, root="data/datasets"):
        self.categorical_cols = categorical_cols
        folder = os.path.join(root, data_name)
        files = {k: os.path.join(folder, k + ".pt") for k in ("data", "min", "max")}
        if not all(os.path.exists(f) for f in files.values()):
            if load_path is None or not load_path.endswith(".pt") or not os.path.exists(load_path):
                raise AssertionError(f"no cache under {folder}: load_path must name an existing raw .pt tensor (got {load_path!r})")
            print(f"no cache under {folder}: building it from {load_path}")
            self._build(load_path, files, cut_length)
        self.data = _load_mapped(files["data"])[:, :, :cut_length]
        self.min = torch.load(files["min"], weights_only=True)
        self.max = torch.load(files["max"], weights_only=True)
        print(f"Data dimension: {self.data.shape}")

    def _build(self, load_path, files, cut_length):
        raw = _load_mapped(load_path)
        assert raw.dim() == 3, f"expected an (N, C, L) tensor in {load_path}, got shape {tuple(raw.shape)}"
        N, C, L = raw.shape
        keep = list(self.categorical_cols) if self.categorical_cols is not None else []
        if keep:
            assert not torch.isnan(raw[:, keep, :]).any(), "indicator columns contain NaNs"
        stats = _ChannelStats(C)
        for i in range(0, N, self.CHUNK):
            stats.add(raw[i:i + self.CHUNK])
        mean = torch.from_numpy(stats.mean).float()[None, :, None]
        low = torch.from_numpy(stats.low).float()[None, :, None]
        span = torch.from_numpy(stats.high - stats.low).float()[None, :, None]
        os.makedirs(os.path.dirname(files["data"]), exist_ok=True)
        Lc = min(L, cut_length)
        stage_path = files["data"] + ".stage"
        stage = torch.from_file(stage_path, shared=True, size=N * C * Lc, dtype=torch.float32).view(N, C, Lc)
        for i in range(0, N, self.CHUNK):
            x = raw[i:i + self.CHUNK, :, :Lc].float()
            x = torch.where(torch.isnan(x), mean, x)
            y = (x - low) / span
            if keep:
                y[:, keep, :] = x[:, keep, :]
            rest = [c for c in range(C) if c not in keep]
            # as normalize_data (utils.py:143 of the reference): NaNs -- a constant channel gives 0 / 0 -- are left out of the check
            inside = y[:, rest][~torch.isnan(y[:, rest])]
            assert bool(((inside >= 0) & (inside <= 1)).all()), "failed to normalize data"
            stage[i:i + self.CHUNK] = y
        torch.save(stage, files["data"])
        del stage
        os.remove(stage_path)
        torch.save(low, files["min"])
        torch.save(low + span, files["max"])
        print(f"cache written to {os.path.dirname(files['data'])}")

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, idx):
        return self.data[idx].clone()

    def __getitems__(self, indices):
        """One gather for the whole batch (sorted, so a memory-mapped cache is read front to back)."""
        order = torch.as_tensor(indices)
        srt, back = torch.sort(order)
        return self.data[srt][torch.argsort(back)] if len(order) > 1 else self.data[order]

    @property
    def channels(self):
        return self.data.shape[1]

    @property
    def seq_length(self):
        return self.data.shape[2]


class TensorSeriesDataset(Dataset):
    """An in-memory (N, C, L) tensor with the TimeSeriesDataset interface (synthetic runs, tests)."""

    def __init__(self, data, categorical_cols, min=None, max=None):
        self.data = data
        self.categorical_cols = categorical_cols
        C = data.shape[1]
        self.min = torch.zeros(1, C, 1) if min is None else min
        self.max = torch.ones(1, C, 1) if max is None else max

    def __len__(self):
        return self.data.shape[0]

    def __getitem__(self, idx):
        return self.data[idx].clone()

    def __getitems__(self, indices):
        return self.data[torch.as_tensor(indices)]

    @property
    def channels(self):
        return self.data.shape[1]

    @property
    def seq_length(self):
        return self.data.shape[2]


def reverse_to_nonan_indicator(cum):
    if isinstance(cum, torch.Tensor):
        return torch.cat([cum[:1], cum[1:] - cum[:-1]])
    if isinstance(cum, np.ndarray):
        return np.concatenate([cum[:1], cum[1:] - cum[:-1]])
    raise ValueError()


def exists(x):
    return x is not None


def default(val, d):
    if exists(val):
        return val
    return d() if callable(d) else d


def extract(a, t, x_shape):
    """utils.py:70-73 (kept for API parity; the kernels do this lookup themselves)."""
    b, *_ = t.shape
    return a.gather(-1, t).reshape(b, *((1,) * (len(x_shape) - 1)))


def normalize_to_neg_one_to_one(sample):
    return sample * 2 - 1


def unnormalize_to_zero_to_one(t):
    return (t + 1) * 0.5


def identity(t, *args, **kwargs):
    return t


def cycle(dl):
    while True:
        for data in dl:
            yield data


def check_is_integer(var):
    assert isinstance(var, int), "must be integer!"


def times_two_if_bidirectional(integer, bidirectional):
    return integer * 2 if bidirectional else integer


def normalize_data(data, categorical_cols, eps=0):
    """utils.py:118-148 for an in-memory tensor: (data, min, max) with every channel but the indicator ones mapped to
    [0, 1] by its own min / max over (samples, time); NaNs stay NaN and are ignored by the statistics."""
    x = data.double()
    keep = list(categorical_cols) if categorical_cols is not None else []
    if keep:
        assert not torch.isnan(x[:, keep, :]).any(), "indicator columns contain NaNs"
    filled_hi = torch.where(torch.isnan(x), torch.full_like(x, -float("inf")), x)
    filled_lo = torch.where(torch.isnan(x), torch.full_like(x, float("inf")), x)
    low = filled_lo.amin(dim=(0, 2))[None, :, None]
    high = filled_hi.amax(dim=(0, 2))[None, :, None]
    out = (x - low + eps) / (high - low + 2 * eps)
    rest = [c for c in range(x.shape[1]) if c not in keep]
    inside = out[:, rest][~torch.isnan(out[:, rest])]
    assert bool(((inside >= 0) & (inside <= 1)).all()), "failed to normalize data"
    if keep:
        out[:, keep, :] = x[:, keep, :]
    return out.float(), low.float(), high.float()


def reverse_normalize(data, min, max):
    return data * (max - min) + min


def replace_nan_with_mean(data):
    """NaN -> the channel's mean over (samples, time) of its non-NaN entries (utils.py:153-157)."""
    return replace_nan_with(data, data.nanmean(dim=(0, 2))[None, :, None])


def replace_nan_with(data, value):
    """NaN -> value (a number, or a tensor that broadcasts against data) (utils.py:159-162)."""
    value = torch.as_tensor(value, dtype=data.dtype)
    out = torch.where(torch.isnan(data), value, data)
    if torch.isnan(out).any():
        raise AssertionError("failed to replace NaNs!")
    return out

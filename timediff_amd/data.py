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


class TimeSeriesDataset(Dataset):
    """utils.py:10-51: (N, C, L) tensor cached under data/datasets/<name>/{data,min,max}.pt; built from a
    raw .pt (NaN -> channel mean, min-max normalisation of the non-indicator channels) on first use."""

    def __init__(self, data_name, categorical_cols, cut_length, load_path=None, root="data/datasets"):
        path = f"{root}/{data_name}"
        self.categorical_cols = categorical_cols
        files = [f"{path}/{n}.pt" for n in ("data", "min", "max")]
        if all(os.path.exists(f) for f in files):
            self.data = torch.load(files[0], weights_only=True)[:, :, :cut_length]
            self.min = torch.load(files[1], weights_only=True)
            self.max = torch.load(files[2], weights_only=True)
        else:
            print("data not found in datasets directory, will load raw data and save to dataset folder")
            assert load_path is not None, "load_path must be specified if data does not exist in dataset folder"
            assert load_path.endswith(".pt"), "load_path must be a .pt file"
            assert os.path.exists(load_path), "load_path does not exist"
            os.makedirs(path, exist_ok=True)
            data = replace_nan_with_mean(torch.load(load_path, weights_only=True))
            data, self.min, self.max = normalize_data(data, categorical_cols)
            self.data = data[:, :, :cut_length]
            torch.save(self.data, files[0])
            torch.save(self.min, files[1])
            torch.save(self.max, files[2])
            print(f"data saved to {path} folder")
        print(f"Data dimension: {self.data.shape}")

    def __len__(self):
        return len(self.data)

    def __getitem__(self, idx):
        return self.data[idx].clone()

    def __getitems__(self, indices):
        """Batched fetch (torch's DataLoader calls this when present): one gather instead of len(indices) clones."""
        return self.data[torch.as_tensor(indices)]

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

    __len__ = TimeSeriesDataset.__len__
    __getitem__ = TimeSeriesDataset.__getitem__
    __getitems__ = TimeSeriesDataset.__getitems__
    channels = TimeSeriesDataset.channels
    seq_length = TimeSeriesDataset.seq_length


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
    """utils.py:118-148: min-max over (samples, time) per channel; indicator channels are left as they are."""
    data = data.numpy()
    if categorical_cols is not None:
        indicators = data[:, categorical_cols, :].copy()
        assert not np.any(np.isnan(indicators)), "indicator columns contain NaNs"
    mn, mx = np.nanmin(data, axis=(0, 2)), np.nanmax(data, axis=(0, 2))
    mn, mx = mn[None, :, None], mx[None, :, None]
    data = (data - mn + eps) / (mx - mn + 2 * eps)
    ok = data[~np.isnan(data)]
    assert np.all((ok >= 0) & (ok <= 1)), "failed to normalize data"
    if categorical_cols is not None:
        data[:, categorical_cols, :] = indicators
    return torch.tensor(data).float(), torch.tensor(mn).float(), torch.tensor(mx).float()


def reverse_normalize(data, min, max):
    return data * (max - min) + min


def replace_nan_with_mean(data):
    mean = data.nanmean(dim=(0, 2))[None, :, None]
    data = torch.where(torch.isnan(data), mean, data)
    assert torch.all(~torch.isnan(data)), "failed to replace NaNs!"
    return data


def replace_nan_with(data, value):
    data = torch.where(torch.isnan(data), torch.tensor(value), data)
    assert torch.all(~torch.isnan(data)), "failed to replace NaNs!"
    return data

"""CPU tests of the host side: CLI surface, module/buffer layout, EMA schedule, patient sharding and the
world_size-2 (gloo) multi-process paths.  No kernel is launched here."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden, sub


def test_cli_defaults_match_reference():
    """etdiff_train.py:32-148 of the reference: flag names and defaults."""
    import etdiff_train as cli
    a = cli.parse_arguments(["--check_point_path", "x.pt"])
    want = dict(data_name="eicu", cut_length=272, hidden=256, num_layers=3, model="lstm", bidirectional=False,
                batch_size=32, learning_rate=8e-5, embed_dim=64, time_dim=256, dropout=0, num_steps=700000,
                gradient_accumulate_every=2, ema_decay=0.995, timesteps=1000, auto_normalize=True, record_every=10,
                wandb=False, eval=False, seed=2023, loss_lambda=0.8, diff_type="mixed", dim=64)
    for k, v in want.items():
        assert getattr(a, k) == v, k
    assert cli.COLUMNS_DICT["eicu"] == {"numerical": [0, 2, 4, 6], "categorical": [1, 3, 5, 7, 8],
                                        "categorical_num_classes": [2] * 5}
    assert cli.COLUMNS_DICT["mimiciii"]["categorical"] == [1, 3, 5, 7, 9, 11, 13, 14]
    assert cli.model_path_to_sample_path("results/models/a.pt") == "results/samples/a_samples.npy"
    with pytest.raises(SystemExit):
        cli.parse_arguments([])            # --check_point_path is required
    with pytest.raises(SystemExit):
        cli.parse_arguments(["--check_point_path", "x", "--batch_size", "0"])


def test_reference_import_paths_and_state_dict_layout():
    """The names etdiff_train.py imports, and the checkpoint's `diff_model` keys in the reference's order."""
    from models.ETDiff.blocks import RNN, EncoderDecoderRNN, NeuralCDE  # noqa: F401
    from models.ETDiff.mixed_diffusion import MixedDiffusion
    from models.ETDiff.utils import TimeSeriesDataset, cosine_beta_schedule, extract  # noqa: F401
    g = load_golden("train_steps_eicu")
    L, H, N, T, B, ga = [int(v) for v in g["cfg"]]
    rnn = RNN(input_channels=14, hidden_channels=H, output_channels=14, layers=N, bidirectional=True)
    diff = MixedDiffusion(rnn, seq_length=L, channels=14, numerical_features_indices=list(g["num_idx"]),
                          categorical_features_indices=list(g["cat_idx"]), categorical_num_classes=list(g["K"]),
                          timesteps=T, loss_lambda=0.8)
    ref = sub(g, "init.")
    sd = diff.state_dict()
    assert list(sd.keys()) == list(ref.keys())
    for k, v in ref.items():
        assert tuple(sd[k].shape) == v.shape and str(sd[k].dtype).endswith(str(v.dtype)), k
        if not k.startswith("model."):
            np.testing.assert_array_equal(sd[k].numpy(), v, err_msg=k)     # schedule buffers: bit-exact
    assert diff.self_condition is False and diff.num_timesteps == T and diff.seq_length == L
    with pytest.raises(ValueError):
        MixedDiffusion(rnn, L, 14, [0], [1], [2], beta_schedule="sigmoid")
    with pytest.raises(ValueError):
        RNN(14, 8, 14, 1, model="rnn")
    with pytest.raises(RuntimeError):
        diff(torch.zeros(2, 9, L))          # no CPU fallback: the module must be on the GPU
    with pytest.raises(AssertionError):
        diff(torch.zeros(2, 9, L + 1))      # mixed_diffusion.py:405


def test_ema_schedule():
    """The warm-up rule documented in timediff_amd/et_diff.py (ema_pytorch defaults; parity unpinned)."""
    from timediff_amd.et_diff import EMA
    ema = EMA(torch.nn.Linear(2, 2), beta=0.995, update_every=10)
    plan = [ema.plan_update() for _ in range(130)]
    assert all(m == 0 for i, (m, _) in enumerate(plan, 1) if i % 10)
    assert [plan[i - 1][0] for i in (10, 50, 100)] == [1, 1, 1]          # copies until step 100
    assert plan[109][0] == 2 and abs(plan[109][1] - (1 - (1 + 9) ** (-2 / 3))) < 1e-12
    assert plan[129][0] == 2 and plan[129][1] <= 0.995
    assert ema.current_decay(10 ** 7) == 0.995
    assert set(ema.state_dict()) == {"initted", "step", "ema_model.weight", "ema_model.bias"}


def test_patient_shard_partition():
    from timediff_amd.parallel import patient_shard
    for total in (0, 1, 7, 20000, 1000000):
        for world in (1, 2, 3, 8):
            spans = [patient_shard(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class _FakeSampler:
    """Stands in for MixedDiffusion.sample: the value of a row depends on (seed, global patient id) only."""
    channels, seq_length, categorical_features_indices = 3, 4, []

    def __init__(self):
        self.calls = []

    def eval(self):
        return self

    @staticmethod
    def expected(total, seed):
        ids = np.arange(total, dtype=np.float32)[:, None, None]
        return (ids * 0.25 + seed + np.arange(3, dtype=np.float32)[None, :, None] * 0.5 + np.zeros((1, 1, 4), np.float32))

    def sample(self, batch_size, seed=None, patient0=0, to_cpu=True):
        self.calls.append(batch_size)
        return torch.from_numpy(self.expected(patient0 + batch_size, seed)[patient0:])


def _worker(rank, world, port, out, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    sys.path.insert(0, ROOT)
    import etdiff_train as cli
    from timediff_amd.parallel import all_reduce_mean_, patient_shard
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # data-parallel gradient exchange: mean of the ranks' flat buffers
        flat = torch.arange(10, dtype=torch.float32) * (rank + 1)
        all_reduce_mean_(flat)
        assert torch.allclose(flat, torch.arange(10, dtype=torch.float32) * (sum(range(1, world + 1)) / world))
        # patient-sharded sampling through the streaming writer: every rank draws its slice of the patient range
        # (values keyed by the GLOBAL patient id, like the Philox streams) and writes its rows of the .npy itself
        model = _FakeSampler()
        mn, mx = np.zeros((1, 3, 1), np.float32), np.full((1, 3, 1), 2.0, np.float32)
        os.environ["TIMEDIFF_SAMPLE_CHUNK"] = "4"
        path = cli.collect_samples_to_file(model, 23, 2, mn, mx, out_path, rank, world, seed=5)
        a, b = patient_shard(22, rank, world)
        assert sum(model.calls) == b - a and max(model.calls) <= 4
        got = np.load(path)
        assert got.shape == (22, 3, 4) and got.dtype == np.float32
        np.testing.assert_array_equal(got, _FakeSampler.expected(22, 5) * 2.0)
        dist.barrier()
        out.put((rank, "ok"))
    finally:
        dist.destroy_process_group()


def test_world_size_2_gloo(tmp_path):
    """Two gloo ranks: the gradient mean, and the sharded sample file equals the single-rank file for the same
    (seed, patient ids)."""
    import etdiff_train as cli
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    out_path = str(tmp_path / "two" / "samples.npy")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out, out_path)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(out.get(timeout=5) for _ in range(2)) == [(0, "ok"), (1, "ok")]
    single = cli.collect_samples_to_file(_FakeSampler(), 23, 2, np.zeros((1, 3, 1), np.float32), np.full((1, 3, 1), 2.0, np.float32),
                                         str(tmp_path / "one" / "samples.npy"), 0, 1, seed=5)
    np.testing.assert_array_equal(np.load(single), np.load(out_path))


def test_sample_writer_squeezes_like_the_reference(tmp_path):
    """torch.cat(samples).squeeze() (etdiff_train.py:163): size-1 axes are dropped from the saved array."""
    from timediff_amd.parallel import SampleWriter
    w = SampleWriter(str(tmp_path / "s.npy"), 6, 1, 5)
    w.put(0, np.arange(30, dtype=np.float32).reshape(6, 1, 5))
    w.close()
    got = np.load(tmp_path / "s.npy")
    assert got.shape == (6, 5)
    np.testing.assert_array_equal(got, np.arange(30, dtype=np.float32).reshape(6, 5))
    with pytest.raises(AssertionError):
        SampleWriter(str(tmp_path / "t.npy"), 4, 2, 3).put(3, np.zeros((2, 2, 3), np.float32))


def test_dataset_cache_roundtrip(tmp_path):
    """TimeSeriesDataset (utils.py:10-51): NaN -> channel mean, min-max of the non-indicator channels, cache files."""
    from timediff_amd.data import TimeSeriesDataset, reverse_normalize
    torch.manual_seed(0)
    raw = torch.rand(20, 4, 9) * 10 - 3
    raw[:, 1, :] = (torch.rand(20, 9) > 0.5).float()
    raw[3, 0, 2] = float("nan")
    torch.save(raw, tmp_path / "raw.pt")
    ds = TimeSeriesDataset("toy", [1], 8, load_path=str(tmp_path / "raw.pt"), root=str(tmp_path / "datasets"))
    assert ds.data.shape == (20, 4, 8) and ds.channels == 4 and ds.seq_length == 8 and len(ds) == 20
    assert torch.equal(ds.data[:, 1], raw[:, 1, :8])
    num = ds.data[:, [0, 2, 3]]
    assert num.min() >= 0 and num.max() <= 1 and not torch.isnan(ds.data).any()
    again = TimeSeriesDataset("toy", [1], 8, root=str(tmp_path / "datasets"))
    assert torch.equal(again.data, ds.data) and torch.equal(again.min, ds.min)
    back = reverse_normalize(ds.data.numpy(), ds.min.numpy()[..., :1], ds.max.numpy()[..., :1])
    ok = ~torch.isnan(raw[:, :, :8]).numpy()
    np.testing.assert_allclose(back[:, [0, 2, 3]][ok[:, [0, 2, 3]]], raw[:, [0, 2, 3], :8].numpy()[ok[:, [0, 2, 3]]], rtol=1e-4,
                               atol=1e-4)


def test_dataset_streamed_build_equals_in_memory(tmp_path, monkeypatch):
    """The cache built in chunks of samples (two streamed passes, memory-mapped staging) equals fill-then-normalise on
    the whole tensor (utils.py:24-27: replace_nan_with_mean, then normalize_data), and batches come back in request order."""
    from timediff_amd import data as D
    torch.manual_seed(3)
    raw = torch.randn(53, 5, 11) * 4 + 1
    raw[:, 3, :] = (torch.rand(53, 11) > 0.3).float()
    raw[torch.rand(53, 5, 11) < 0.1] = float("nan")
    raw[:, 3, :] = torch.nan_to_num(raw[:, 3, :], nan=1.0)
    torch.save(raw, tmp_path / "raw.pt")
    monkeypatch.setattr(D.TimeSeriesDataset, "CHUNK", 7)
    ds = D.TimeSeriesDataset("toy", [3], 9, load_path=str(tmp_path / "raw.pt"), root=str(tmp_path / "datasets"))
    want, lo, hi = D.normalize_data(D.replace_nan_with_mean(raw), [3])
    assert torch.allclose(ds.data, want[:, :, :9], rtol=0, atol=2e-6)
    assert torch.allclose(ds.min, lo, atol=1e-6) and torch.allclose(ds.max, hi, atol=1e-6)
    assert not os.path.exists(tmp_path / "datasets" / "toy" / "data.pt.stage")
    idx = [40, 3, 17, 3, 52, 0]
    assert torch.equal(ds.__getitems__(idx), ds.data[torch.tensor(idx)])
    with pytest.raises(AssertionError):
        D.TimeSeriesDataset("other", [3], 9, load_path=str(tmp_path / "missing.pt"), root=str(tmp_path / "datasets"))


def test_dataset_build_with_a_constant_channel(tmp_path, monkeypatch):
    """A channel whose max equals its min normalises to 0 / 0 = NaN; the reference's normalize_data (utils.py:143) leaves
    NaNs out of its range check and builds the cache, and so does the streamed build -- with the same values."""
    from timediff_amd import data as D
    torch.manual_seed(4)
    raw = torch.randn(20, 4, 6)
    raw[:, 2, :] = 3.5
    torch.save(raw, tmp_path / "raw.pt")
    monkeypatch.setattr(D.TimeSeriesDataset, "CHUNK", 6)
    ds = D.TimeSeriesDataset("const", None, 6, load_path=str(tmp_path / "raw.pt"), root=str(tmp_path / "datasets"))
    want, lo, hi = D.normalize_data(D.replace_nan_with_mean(raw), None)
    assert torch.isnan(ds.data[:, 2]).all() and torch.isnan(want[:, 2]).all()
    rest = [0, 1, 3]
    assert torch.allclose(ds.data[:, rest], want[:, rest], rtol=0, atol=2e-6)
    assert torch.allclose(ds.min, lo, atol=1e-6) and torch.allclose(ds.max, hi, atol=1e-6)


def test_layer_slices_and_uncovered_ranges():
    """The flat gradient buffer's per-layer all-reduce slices (ETDiff._layer_reduce) and what is left for the end."""
    import torch.nn as nn
    from timediff_amd.parallel import layer_slices, uncovered_ranges
    m = nn.ModuleDict({"model": nn.ModuleDict({"pre": nn.Linear(3, 2), "rnn": nn.ModuleDict({"rnn": nn.LSTM(5, 4, 3, bidirectional=True)}),
                                               "fc": nn.Linear(8, 5)})})
    names = [n for n, _ in m.named_parameters()]
    sizes = [p.numel() for p in m.parameters()]
    sl = layer_slices(names, sizes, 3)
    per_layer = [2 * (16 * (5 if l == 0 else 8) + 16 * 4 + 32) for l in range(3)]
    assert sl[0][0] == 3 * 2 + 2 and [b - a for a, b in sl] == per_layer
    assert sl[0][1] == sl[1][0] and sl[1][1] == sl[2][0]
    n = sum(sizes)
    assert uncovered_ranges(n, [sl[2], sl[0], sl[1]]) == [(0, sl[0][0]), (sl[2][1], n)]
    assert uncovered_ranges(n, []) == [(0, n)]
    assert uncovered_ranges(10, [(0, 4), (4, 10)]) == []


def test_collect_samples_fuses_chains(monkeypatch):
    """collect_samples (etdiff_train.py:150-166): num_samples // batch_size chains of batch_size patients, drawn in
    GPU-filling chunks; same count, order of concatenation and de-normalisation as the reference."""
    import etdiff_train as cli

    class Fake:
        def __init__(self):
            self.calls = []

        def eval(self):
            return self

        def sample(self, batch_size):
            self.calls.append(batch_size)
            return torch.full((batch_size, 3, 4), 0.5)

    m = Fake()
    out = cli.collect_samples(m, 1050, batch_size=100, min=np.zeros((1, 3, 1)), max=np.full((1, 3, 1), 2.0))
    assert out.shape == (1000, 3, 4) and m.calls == [1000] and np.allclose(out, 1.0)
    monkeypatch.setenv("TIMEDIFF_SAMPLE_CHUNK", "300")
    m = Fake()
    out = cli.collect_samples(m, 1000, batch_size=100)
    assert out.shape == (1000, 3, 4) and m.calls == [300, 300, 300, 100]

"""GPU parity of the training step behind the reference's module interface (models/ETDiff/et_diff.py:226-261):
MixedDiffusion.forward -> loss.backward() -> td_grad_norm / td_clip_adam_ema, against the golden vectors of
two optimizer steps of the reference loop, plus the fused optimizer kernels against the oracle, and the
checkpoint layout."""
import numpy as np
import pytest
import torch

from conftest import load_golden, sub
from oracle import oracle as orc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    return torch.device("cuda:0")


def _build(g, dev, ga, lr, tmp_path, T, L, H, N, fuse=True):
    from models.ETDiff.blocks import RNN
    from models.ETDiff.et_diff import ETDiff
    from models.ETDiff.mixed_diffusion import MixedDiffusion
    from timediff_amd.data import TensorSeriesDataset
    num, cat, K = list(g["num_idx"]), list(g["cat_idx"]), list(g["K"])
    width = len(num) + int(sum(K))
    rnn = RNN(input_channels=width, hidden_channels=H, output_channels=width, layers=N, model="lstm", bidirectional=True)
    diff = MixedDiffusion(rnn, seq_length=L, channels=width, numerical_features_indices=num,
                          categorical_features_indices=cat, categorical_num_classes=K, timesteps=T, loss_lambda=0.8)
    missing = diff.load_state_dict({k: torch.from_numpy(v) for k, v in sub(g, "init.").items()})
    assert not missing.missing_keys and not missing.unexpected_keys
    ds = TensorSeriesDataset(torch.from_numpy(g["data"].reshape(-1, *g["data"].shape[2:])), cat)
    et = ETDiff(diff, ds, train_batch_size=g["data"].shape[1], gradient_accumulate_every=ga, diff_lr=lr,
                diff_num_steps=2, ema_update_every=1, check_point_path=str(tmp_path / "ckpt.pt"), fuse_accumulation=fuse)
    return et


@pytest.mark.parametrize("fuse", [False, True])
def test_train_steps_against_reference(fuse, dev, tmp_path):
    """Two optimizer steps of the reference loop (gradient_accumulate_every = 2), micro-batch by micro-batch as the
    reference runs them (fuse = False) and with the micro-batches of a step concatenated into one launch (the default):
    the same golden losses, gradient norms and weights."""
    g = load_golden("train_steps_eicu")
    L, H, N, T, B, ga = [int(v) for v in g["cfg"]]
    lr = float(g["lr"])
    et = _build(g, dev, ga, lr, tmp_path, T, L, H, N, fuse)
    # feed the reference's batches and its recorded draws (t, noise, uniforms) in order
    batches = iter([torch.from_numpy(b.copy()) for b in g["data"]])
    per_call = ga if fuse else 1       # the fused step draws for the concatenated batch
    draws = iter([{k: torch.cat([torch.from_numpy(g[k][i + j]) for j in range(per_call)]).to(dev) for k in ("t", "noise", "uniforms")}
                  for i in range(0, len(g["t"]), per_call)])
    et.dl = batches
    inner = et.model.forward
    et.model.forward = lambda sample: inner(sample, **next(draws))
    for step in range(2):
        loss = et.train_step().item()
        np.testing.assert_allclose(loss, g["total_loss"][step], rtol=2e-2)
        np.testing.assert_allclose(et._norm.item(), g["grad_norm"][step], rtol=5e-2)
        sd = {k: v.detach().cpu().numpy() for k, v in et.model.state_dict().items()}
        for k, ref in sub(g, f"step{step}.").items():
            err = np.abs(sd[k] - ref)
            # Adam's first steps move every weight by ~lr * sign(g); bf16 gradients flip the sign only where g ~ 0
            assert err.max() <= 2.2 * lr * (step + 1), (k, err.max())
            assert (err > 0.3 * lr).mean() < 0.08, (k, (err > 0.3 * lr).mean())
    # the EMA wrapper tracked the online weights (copy during warm-up) in its own flat buffer
    for p, q in zip(et.model.parameters(), et.ema.ema_model.parameters()):
        assert torch.equal(p, q) and p.data_ptr() != q.data_ptr()

    # checkpoint: the reference's dict layout, loadable by stock torch objects
    et.save_weights()
    ck = torch.load(et.check_point_path, map_location="cpu", weights_only=False)
    assert sorted(ck) == ["diff_model", "diff_step", "ema", "opt", "scaler"] and ck["diff_step"] == 2
    ref_keys = [k[len("init."):] for k in g if k.startswith("init.")]
    assert list(ck["diff_model"].keys()) == ref_keys
    assert all(k.startswith("ema_model.") or k in ("initted", "step") for k in ck["ema"])
    params = [torch.nn.Parameter(v.clone()) for k, v in ck["diff_model"].items() if k.startswith("model.")]
    torch.optim.Adam(params, lr=lr, betas=(0.9, 0.99)).load_state_dict(ck["opt"])
    # resume: a fresh trainer restores weights, step count and Adam moments
    et2 = _build(g, dev, ga, lr, tmp_path, T, L, H, N, fuse)
    et2.load(et.check_point_path)
    assert et2.step == 2 and et2.opt_step == 2
    assert torch.equal(et2.flat_param, et.flat_param) and torch.equal(et2.exp_avg_sq, et.exp_avg_sq)
    assert torch.equal(et2.flat_ema, et.flat_ema)
    # the sampler of the EMA model runs from the trained weights
    out = et.ema.ema_model.sample(batch_size=3)
    assert out.shape == (3, 9, L) and out.device.type == "cpu" and torch.isfinite(out).all()
    assert set(np.unique(out[:, list(g["cat_idx"])].numpy())) <= {0.0, 1.0}


@pytest.mark.parametrize("n", [1, 1000, 4064782])
def test_fused_optimizer_against_oracle(n, dev):
    """td_grad_norm + td_clip_adam_ema on a flat buffer == clip_grad_norm_(1.0) + Adam + EMA lerp."""
    import ctypes as C
    from timediff_amd._lib import check, lib, ptr
    rng = np.random.default_rng(n)
    p = rng.standard_normal(n).astype(np.float32)
    g = (rng.standard_normal(n) * (3.0 / np.sqrt(n))).astype(np.float32)     # norm ~ 3: clipping is active
    m = (rng.standard_normal(n) * 0.01).astype(np.float32)
    v = (rng.random(n) * 1e-4).astype(np.float32)
    e = rng.standard_normal(n).astype(np.float32)
    tp, tg, tm, tv, te = [torch.from_numpy(a.copy()).to(dev) for a in (p, g, m, v, e)]
    norm = torch.zeros(1, device=dev)
    ws = torch.empty(lib().td_grad_norm_workspace_bytes(), dtype=torch.uint8, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    check(lib().td_grad_norm(ptr(tg), n, ptr(norm), ptr(ws), st))
    want_norm = float(np.sqrt((g.astype(np.float64) ** 2).sum()))
    np.testing.assert_allclose(norm.item(), want_norm, rtol=1e-6)
    step, lr, b1, b2, eps, decay = 7, 8e-5, 0.9, 0.99, 1e-8, 0.995
    check(lib().td_clip_adam_ema(ptr(tp), ptr(tg), ptr(tm), ptr(tv), ptr(te), n, ptr(norm), 1.0, lr, b1, b2, eps, step, 2,
                                 decay, st))
    params, grads, ms, vs = {"w": p.copy()}, {"w": g.copy()}, {"w": m.copy()}, {"w": v.copy()}
    onorm = orc.clip_and_adam(params, grads, ms, vs, step, lr)
    np.testing.assert_allclose(onorm, want_norm, rtol=1e-5)
    np.testing.assert_allclose(tp.cpu().numpy(), params["w"], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(tm.cpu().numpy(), ms["w"], rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(tv.cpu().numpy(), vs["w"], rtol=1e-5, atol=1e-12)
    np.testing.assert_allclose(te.cpu().numpy(), decay * e + (1 - decay) * params["w"], rtol=1e-5, atol=1e-6)
    # ema_mode 1 copies, 0 leaves alone; max_norm <= 0 skips the clip
    te0 = te.clone()
    check(lib().td_clip_adam_ema(ptr(tp), ptr(tg), ptr(tm), ptr(tv), ptr(te), n, None, 0.0, lr, b1, b2, eps, step + 1, 0,
                                 decay, st))
    assert torch.equal(te, te0)
    check(lib().td_clip_adam_ema(ptr(tp), ptr(tg), ptr(tm), ptr(tv), ptr(te), n, None, 0.0, lr, b1, b2, eps, step + 2, 1,
                                 decay, st))
    assert torch.equal(te, tp)


def test_gaussian_training_step(dev, tmp_path):
    """--diff_type gaussian: loss.backward() reaches every parameter and a step lowers nothing to NaN."""
    from models.ETDiff.blocks import RNN
    from models.ETDiff.et_diff import ETDiff
    from models.ETDiff.gaussian_diffusion import GaussianDiffusion
    from timediff_amd.data import TensorSeriesDataset
    torch.manual_seed(0)
    rnn = RNN(input_channels=6, hidden_channels=32, output_channels=6, layers=2, model="gru", bidirectional=True)
    diff = GaussianDiffusion(rnn, seq_length=20, channels=6, timesteps=30)
    ds = TensorSeriesDataset(torch.rand(64, 6, 20), None)
    et = ETDiff(diff, ds, train_batch_size=16, gradient_accumulate_every=1, diff_lr=1e-3, diff_num_steps=3,
                check_point_path=str(tmp_path / "g.pt"))
    before = et.flat_param.clone()
    losses = [et.train_step().item() for _ in range(3)]
    assert all(np.isfinite(losses))
    assert (et.flat_param != before).float().mean() > 0.99      # every parameter received a gradient
    out = et.ema.ema_model.sample(batch_size=5)
    assert out.shape == (5, 6, 20) and torch.isfinite(out).all()


def test_checkpoint_written_beside_training(dev, tmp_path):
    """save_weights(block=False): the snapshot is taken in stream order, the file is written by a thread while further
    optimizer steps change the weights; the file holds the weights, EMA and Adam moments of the step it was asked at,
    in the reference's dict layout (et_diff.py:208-216), and a fresh trainer resumes from it."""
    g = load_golden("train_steps_eicu")
    L, H, N, T, B, ga = [int(v) for v in g["cfg"]]
    et = _build(g, dev, ga, 1e-3, tmp_path, T, L, H, N)
    for _ in range(3):
        et.train_step()
    want = {k: getattr(et, k).clone() for k in ("flat_param", "flat_ema", "exp_avg", "exp_avg_sq")}
    step = et.step
    et.save_weights(block=False)
    for _ in range(3):                      # the weights move on while the writer runs
        et.train_step()
    et.wait_for_checkpoint()
    assert not torch.equal(et.flat_param, want["flat_param"])
    ck = torch.load(et.check_point_path, map_location="cpu", weights_only=True)
    assert sorted(ck) == ["diff_model", "diff_step", "ema", "opt", "scaler"] and ck["diff_step"] == step
    assert list(ck["diff_model"].keys()) == list(et.model.state_dict().keys())
    assert list(ck["ema"].keys()) == list(et.ema.state_dict().keys())
    et2 = _build(g, dev, ga, 1e-3, tmp_path, T, L, H, N)
    et2.load(et.check_point_path)
    for k, v in want.items():
        assert torch.equal(getattr(et2, k), v), k
    assert et2.step == step and et2.opt_step == step
    # a second checkpoint reuses the pinned staging buffers; a writer error surfaces at the join
    et.check_point_path = str(tmp_path / "no_such_dir" / "x.pt")
    et.save_weights(block=False)
    with pytest.raises(Exception):
        et.wait_for_checkpoint()


def _static_draws(g, dev, per_call):
    """Device tensors the patched forward reads its (t, noise, uniforms) from: a captured step bakes addresses, not values."""
    return [{k: torch.empty((per_call * g[k].shape[1],) + g[k].shape[2:], dtype=torch.from_numpy(g[k]).dtype, device=dev)
             for k in ("t", "noise", "uniforms")} for _ in range(2)]


@pytest.mark.parametrize("fuse", [False, True])
def test_train_steps_against_reference_graphed(fuse, dev, tmp_path):
    """The same two golden optimizer steps with the whole step captured in a graph at the first call (no warm-up) and
    replayed: the draws are fed through static device tensors that are rewritten between replays."""
    g = load_golden("train_steps_eicu")
    L, H, N, T, B, ga = [int(v) for v in g["cfg"]]
    lr = float(g["lr"])
    et = _build(g, dev, ga, lr, tmp_path, T, L, H, N, fuse)
    assert et._graph_eligible()
    et.graph_warmup = 0
    et.dl = iter([torch.from_numpy(b.copy()) for b in g["data"]])
    per_call = ga if fuse else 1
    calls = ga // per_call                      # forward calls per optimizer step
    static = _static_draws(g, dev, per_call)
    inner, count = et.model.forward, [0]

    def forward(sample):
        j = count[0] % calls
        count[0] += 1
        return inner(sample, **static[j])
    et.model.forward = forward
    for step in range(2):
        for j in range(calls):
            i0 = (step * calls + j) * per_call
            for k in ("t", "noise", "uniforms"):
                static[j][k].copy_(torch.cat([torch.from_numpy(g[k][i0 + m]) for m in range(per_call)]))
        loss = et.train_step().item()
        assert count[0] == calls, "the step must have been captured once and replayed"
        np.testing.assert_allclose(loss, g["total_loss"][step], rtol=2e-2)
        np.testing.assert_allclose(et._norm.item(), g["grad_norm"][step], rtol=5e-2)
        sd = {k: v.detach().cpu().numpy() for k, v in et.model.state_dict().items()}
        for k, ref in sub(g, f"step{step}.").items():
            err = np.abs(sd[k] - ref)
            assert err.max() <= 2.2 * lr * (step + 1), (k, err.max())
            assert (err > 0.3 * lr).mean() < 0.08, (k, (err > 0.3 * lr).mean())
    assert et.step == 2 and et.opt_step == 2 and len(et._graphs) == 1
    assert float(et.flat_grad.abs().max()) == 0.0
    for p, q in zip(et.model.parameters(), et.ema.ema_model.parameters()):
        assert torch.equal(p, q) and p.data_ptr() != q.data_ptr()
    # the denoiser's weight cache was marked stale: the EMA model samples from the trained weights
    out = et.ema.ema_model.sample(batch_size=3)
    assert out.shape == (3, 9, L) and torch.isfinite(out).all()


def test_graphed_steps_equal_launch_by_launch(dev, tmp_path):
    """Six optimizer steps, noise from the in-kernel generator: a trainer that replays its captured step (seed read from the
    device record) and one that launches kernel by kernel (seed as a kernel argument) draw the same seeds from torch's CPU
    generator and must end at the same weights, Adam moments and EMA (ema_update_every = 1, lerp from step 1 on)."""
    g = load_golden("train_steps_eicu")
    L, H, N, T, B, ga = [int(v) for v in g["cfg"]]
    lr = 1e-3
    data = torch.from_numpy(g["data"].reshape(-1, *g["data"].shape[2:]).copy())
    rows = torch.randint(0, data.shape[0], (6, 2 * B), generator=torch.Generator().manual_seed(5))
    ts = torch.randint(0, T, (6, 2 * B), generator=torch.Generator().manual_seed(6))
    ends = []
    for graphed in (False, True):
        et = _build(g, dev, ga, lr, tmp_path, T, L, H, N, True)
        et.graph_steps, et.graph_warmup = graphed, 1
        et.ema.update_after_step = 0        # decay > 0 from the second update on: the lerp branch of the record runs
        et.dl = iter([data[r].clone() for r in rows.reshape(12, B)])
        t_dev = torch.empty(2 * B, dtype=torch.int64, device=dev)
        inner = et.model.forward
        et.model.forward = lambda sample, inner=inner, t_dev=t_dev: inner(sample, t=t_dev)
        torch.manual_seed(77)
        losses = []
        for step in range(6):
            t_dev.copy_(ts[step])
            losses.append(et.train_step().item())
        assert (len(et._graphs) == 1 and "graph" in next(iter(et._graphs.values()))) == graphed
        ends.append((losses, et.flat_param.clone(), et.exp_avg.clone(), et.exp_avg_sq.clone(), et.flat_ema.clone(), et.opt_step))
    (la, pa, ma, va, ea, sa), (lb, pb, mb, vb, eb, sb) = ends
    assert sa == sb == 6
    np.testing.assert_allclose(la, lb, rtol=1e-4)
    # bias gradients are summed with atomics (order varies in the last bits): Adam's update may differ where g ~ 0
    for a, b, name in ((pa, pb, "param"), (ea, eb, "ema")):
        err = (a - b).abs()
        assert float(err.max()) <= 2.2 * lr * 6, name
        assert float((err > 0.05 * lr).float().mean()) < 0.01, (name, float((err > 0.05 * lr).float().mean()))
    assert not torch.equal(pa, ea)
    # the moments follow the weights: where Adam's first steps flipped a near-zero weight's direction the later gradients differ
    # a little (seen: 5 of 110 k entries of exp_avg off by 7e-6), so they are held to 2 % of the largest entry, and to 0.2 % for
    # all but a hundredth of the entries
    for a, b, name in ((ma, mb, "exp_avg"), (va, vb, "exp_avg_sq")):
        err, top = (a - b).abs(), float(a.abs().max())
        assert float(err.max()) <= 2e-2 * top, (name, float(err.max()), top)
        assert float((err > 2e-3 * top).float().mean()) < 0.01, name


def test_step_record_optimizer_equals_host_arguments(dev):
    """td_clip_adam_ema_rec (step size, bias correction, EMA mode / decay from a td_step_record in device memory) is
    bit-identical to td_clip_adam_ema given the same values as arguments, for every EMA mode."""
    import ctypes as C
    from timediff_amd import _lib
    from timediff_amd._lib import check, lib, ptr
    n = 100003
    gen = torch.Generator(device="cpu").manual_seed(3)
    base = [torch.randn(n, generator=gen) for _ in range(5)]
    base[3] = base[3].abs() * 1e-4
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    rec = torch.zeros(C.sizeof(_lib.StepRecord), dtype=torch.uint8, device=dev)
    for mode, step in ((0, 1), (1, 2), (2, 977)):
        a = [t.clone().to(dev) for t in base]
        b = [t.clone().to(dev) for t in base]
        norm = torch.full((1,), 3.0, device=dev)
        check(lib().td_clip_adam_ema(ptr(a[0]), ptr(a[1]), ptr(a[2]), ptr(a[3]), ptr(a[4]), n, ptr(norm), 1.0, 8e-5, 0.9, 0.99,
                                     1e-8, step, mode, 0.9931, st))
        check(lib().td_step_record_set(ptr(rec), 12345, 8e-5, 0.9, 0.99, step, mode, 0.9931, st))
        check(lib().td_clip_adam_ema_rec(ptr(b[0]), ptr(b[1]), ptr(b[2]), ptr(b[3]), ptr(b[4]), n, ptr(norm), 1.0, 0.9, 0.99, 1e-8,
                                         ptr(rec), st))
        for x, y in zip(a, b):
            assert torch.equal(x, y)
        host = _lib.StepRecord.from_buffer_copy(rec.cpu().numpy().tobytes())
        assert host.seed == 12345 and host.ema_mode == mode

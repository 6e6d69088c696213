#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE.

Run in the build container only (needs /root/reference, which never travels to
the GPU box):

    python tests/golden/make_golden.py

The reference's ETDiff modules are imported as they lie in /root/reference
(`torchcde` is an unused import there and is registered as an empty stub).
Nothing of the reference's source is written into this repository: the .npz
files hold inputs, recorded RNG draws, weights and the reference's outputs.

RNG capture: torch.randn / randn_like / rand_like / randint are wrapped so every
draw made by the reference is logged, in order, next to the result it produced
(draw order per SURVEY.md section 8c).
"""
import os
import sys
import types

import numpy as np
import torch

REF = os.environ.get("TIMEDIFF_REFERENCE", "/root/reference")
OUT = os.path.dirname(os.path.abspath(__file__))

sys.modules.setdefault("torchcde", types.ModuleType("torchcde"))
sys.path.insert(0, REF)
import tqdm as _tqdm  # noqa: E402

import models.ETDiff.mixed_diffusion as ref_mixed  # noqa: E402
import models.ETDiff.gaussian_diffusion as ref_gauss  # noqa: E402
import models.ETDiff.blocks as ref_blocks  # noqa: E402
import models.ETDiff.utils as ref_utils  # noqa: E402

# the vectors must come from the reference itself, not from this repository's drop-in modules of the same names
for _m in (ref_mixed, ref_gauss, ref_blocks, ref_utils):
    assert os.path.realpath(_m.__file__).startswith(os.path.realpath(REF) + os.sep), f"{_m.__name__} was imported from {_m.__file__}, not from {REF}"

# silence the progress bar of the reference's sampling loop
ref_mixed.tqdm = lambda it, **kw: it
ref_gauss.tqdm = lambda it, **kw: it

SCHEDULE_KEYS = [
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "log_alpha", "log_1_min_alpha",
    "log_cumprod_alpha", "log_1_min_cumprod_alpha", "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod", "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod", "posterior_variance",
    "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
    "loss_weight",
]

COLUMNS = {
    "eicu": dict(num=[0, 2, 4, 6], cat=[1, 3, 5, 7, 8], K=[2, 2, 2, 2, 2]),
    "mimiciii": dict(num=[0, 2, 4, 6, 8, 10, 12], cat=[1, 3, 5, 7, 9, 11, 13, 14], K=[2] * 8),
    "hirid": dict(num=[0, 1, 2, 3, 4, 5, 6], cat=[7], K=[2]),
    # not a dataset of the reference: exercises K_i != 2 and sum(K) != 2*n_cat
    "k3mix": dict(num=[0, 3], cat=[1, 2, 4], K=[3, 2, 4]),
}


class DrawLog:
    """Wraps torch's RNG entry points used by the reference and records each draw."""

    NAMES = ("randn", "randn_like", "rand_like", "randint")

    def __init__(self):
        self.draws = []
        self._orig = {}

    def __enter__(self):
        for name in self.NAMES:
            orig = getattr(torch, name)
            self._orig[name] = orig

            def wrapped(*a, _orig=orig, _name=name, **kw):
                out = _orig(*a, **kw)
                self.draws.append((_name, out.detach().clone()))
                return out

            setattr(torch, name, wrapped)
        return self

    def __exit__(self, *exc):
        for name, orig in self._orig.items():
            setattr(torch, name, orig)
        return False


def np32(t):
    return t.detach().cpu().numpy().copy()      # copy: state_dict tensors alias live parameters


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print(f"{name}.npz  {os.path.getsize(path) / 1024:.1f} KiB")


def state_arrays(module, prefix="sd."):
    return {prefix + k: np32(v) for k, v in module.state_dict().items()}


def make_data(B, cols, L, gen):
    C = len(cols["num"]) + len(cols["cat"])
    x = torch.rand(B, C, L, generator=gen)
    for i, k in zip(cols["cat"], cols["K"]):
        x[:, i, :] = torch.randint(0, k, (B, L), generator=gen).float()
    return x


def build(cols, L, H, N, T, model="lstm", bidir=True, schedule="cosine", lam=0.8, seed=0):
    torch.manual_seed(seed)
    c_io = len(cols["num"]) + sum(cols["K"])
    rnn = ref_blocks.RNN(c_io, H, c_io, N, model=model, bidirectional=bidir)
    diff = ref_mixed.MixedDiffusion(
        rnn, seq_length=L, channels=c_io,
        numerical_features_indices=cols["num"], categorical_features_indices=cols["cat"],
        categorical_num_classes=cols["K"], timesteps=T, beta_schedule=schedule, loss_lambda=lam)
    # the default init leaves LayerNorm at (1, 0); perturb so the affine terms are exercised
    with torch.no_grad():
        rnn.rnn.norm.weight.add_(0.1 * torch.randn_like(rnn.rnn.norm.weight))
        rnn.rnn.norm.bias.add_(0.1 * torch.randn_like(rnn.rnn.norm.bias))
    return rnn, diff


# --------------------------------------------------------------------------------------
def gen_schedules():
    for sched, T in (("cosine", 1000), ("cosine", 8), ("linear", 50), ("linear", 1000)):
        rnn = ref_blocks.RNN(3, 4, 3, 1)
        d = ref_mixed.MixedDiffusion(rnn, 4, 3, [0], [1], [2], timesteps=T, beta_schedule=sched)
        sd = d.state_dict()
        arrays = {k: np32(sd[k]) for k in SCHEDULE_KEYS}
        arrays["categorical_num_classes_expanded"] = np32(sd["categorical_num_classes_expanded"])
        arrays["offsets"] = np32(sd["offsets"])
        save(f"schedule_{sched}_T{T}", **arrays)


def gen_rnn_forward():
    cases = [
        ("lstm_bi", dict(model="lstm", bidir=True, C=14, H=32, N=3, B=4, L=24)),
        ("lstm_uni", dict(model="lstm", bidir=False, C=9, H=20, N=2, B=3, L=11)),
        ("gru_bi", dict(model="gru", bidir=True, C=14, H=32, N=3, B=4, L=24)),
        ("gru_uni", dict(model="gru", bidir=False, C=9, H=20, N=2, B=3, L=11)),
        ("lstm_bi_h92", dict(model="lstm", bidir=True, C=23, H=92, N=3, B=5, L=24)),
    ]
    for name, c in cases:
        torch.manual_seed(11)
        rnn = ref_blocks.RNN(c["C"], c["H"], c["C"], c["N"], model=c["model"], bidirectional=c["bidir"])
        with torch.no_grad():
            rnn.rnn.norm.weight.add_(0.1 * torch.randn_like(rnn.rnn.norm.weight))
            rnn.rnn.norm.bias.add_(0.1 * torch.randn_like(rnn.rnn.norm.bias))
        x = torch.randn(c["B"], c["C"], c["L"])
        # mimic the denoiser's real input: some channels carry {0, log 1e-30}
        x[:, -4:, :] = torch.where(torch.rand(c["B"], 4, c["L"]) > 0.5, 0.0, float(np.log(np.float32(1e-30))))
        time = torch.tensor([0, 1, 499, 999, 37][: c["B"]], dtype=torch.long)
        x.requires_grad_(True)
        temb = rnn.time_emb(time)
        hidden = rnn.rnn.rnn(x.permute(0, 2, 1))[0]
        y = rnn(x, time)
        gy = torch.randn_like(y)
        y.backward(gy)
        arrays = state_arrays(rnn)
        arrays.update({"grad." + k: np32(p.grad) for k, p in rnn.named_parameters()})
        arrays.update(x=np32(x), time=np32(time), temb=np32(temb), hidden=np32(hidden), y=np32(y),
                      grad_y=np32(gy), grad_x=np32(x.grad),
                      cfg=np.array([c["C"], c["H"], c["N"], int(c["bidir"]), int(c["model"] == "gru")]))
        save(f"rnn_{name}", **arrays)


def gen_mixed_loss():
    cases = [
        ("eicu_T50", "eicu", dict(L=24, H=32, N=3, T=50, B=6, t=[0, 1, 2, 10, 30, 49])),
        ("k3mix_T1000", "k3mix", dict(L=16, H=24, N=2, T=1000, B=6, t=[0, 1, 5, 500, 998, 999])),
        ("hirid_lin_T50", "hirid", dict(L=20, H=36, N=3, T=50, B=4, t=[0, 7, 25, 49], schedule="linear")),
    ]
    for name, colname, c in cases:
        cols = COLUMNS[colname]
        rnn, diff = build(cols, c["L"], c["H"], c["N"], c["T"], schedule=c.get("schedule", "cosine"))
        gen = torch.Generator().manual_seed(5)
        data = make_data(c["B"], cols, c["L"], gen)
        t = torch.tensor(c["t"], dtype=torch.long)
        sample = diff.normalize_numericals(data.clone())

        captured = {}
        orig_model_fwd = rnn.forward

        def hooked(x, time, cond=None):
            captured["x_in"] = x.detach().clone()
            out = orig_model_fwd(x, time, cond)
            out.retain_grad()
            captured["model_out"] = out
            return out

        rnn.forward = hooked
        with DrawLog() as log:
            loss = diff.mixed_loss(sample, t)
        loss.backward()
        rnn.forward = orig_model_fwd

        n_num, n_cat = len(cols["num"]), len(cols["cat"])
        assert [d[0] for d in log.draws] == ["randn_like"] + ["rand_like"] * n_cat
        noise = log.draws[0][1]
        uniforms = torch.cat([d[1] for d in log.draws[1:]], dim=1)      # (B, sumK, L)

        # loss components, recomputed through the reference's own methods
        with torch.no_grad():
            mo = captured["model_out"].detach()
            x_in = captured["x_in"]
            log_x_cat = ref_mixed.index_to_log_onehot(sample[:, cols["cat"], :].long(), diff.categorical_num_classes)
            log_x_cat_t = x_in[:, n_num:]
            log_true = diff.cat_q_posterior(log_x_cat, log_x_cat_t, t)
            log_model = diff.cat_p_pred(mo[:, n_num:], log_x_cat_t, t, None)
            Lt = diff.compute_Lt(mo[:, n_num:], log_x_cat, log_x_cat_t, t)
            klp = diff.kl_prior(log_x_cat)
            lg = diff.gauss_loss(mo[:, :n_num], None, None, t, noise)

        arrays = state_arrays(diff)
        arrays.update({"grad." + k: np32(p.grad) for k, p in diff.named_parameters()})
        arrays.update(
            data=np32(data), t=np32(t), noise=np32(noise), uniforms=np32(uniforms),
            x_in=np32(x_in), model_out=np32(mo), grad_model_out=np32(captured["model_out"].grad),
            loss=np32(loss), log_true=np32(log_true), log_model=np32(log_model), Lt=np32(Lt),
            kl_prior=np32(klp), loss_gauss=np32(lg),
            num_idx=np.array(cols["num"]), cat_idx=np.array(cols["cat"]), K=np.array(cols["K"]),
            cfg=np.array([c["L"], c["H"], c["N"], c["T"]]), loss_lambda=np.float32(diff.loss_lambda))
        save(f"mixed_loss_{name}", **arrays)


def gen_reverse_steps():
    """Single reverse updates of a T=1000 chain at hand-picked t, fed arbitrary model outputs."""
    for colname in ("eicu", "k3mix"):
        cols = COLUMNS[colname]
        L, B = 12, 8
        rnn, diff = build(cols, L, 8, 1, 1000)
        n_num, sumK = len(cols["num"]), sum(cols["K"])
        gen = torch.Generator().manual_seed(9)
        arrays = dict(num_idx=np.array(cols["num"]), cat_idx=np.array(cols["cat"]), K=np.array(cols["K"]))
        steps = [999, 998, 990, 500, 100, 10, 2, 1, 0]
        arrays["steps"] = np.array(steps)
        for i in steps:
            scale = 1.0 if i > 10 else 6.0          # confident logits near the end of the chain
            model_out = scale * torch.randn(B, n_num + sumK, L, generator=gen)
            z = torch.randn(B, n_num, L, generator=gen)
            idx = torch.stack([torch.randint(0, k, (B, L), generator=gen) for k in cols["K"]], dim=1)
            log_z = ref_mixed.index_to_log_onehot(idx, diff.categorical_num_classes)
            t = torch.full((B,), i, dtype=torch.long)
            with DrawLog() as log:
                z_new, x0 = diff.gauss_p_sample(model_out[:, :n_num], z.clone(), t, i, clip_denoised=True)
                log_theta = diff.cat_p_pred(model_out[:, n_num:], log_z, t, None)
                # same draws as cat_p_sample would make after cat_p_pred
                log_z_new = diff.cat_log_sample_categorical(log_theta)
            kinds = [d[0] for d in log.draws]
            assert kinds == (["randn_like"] if i > 0 else []) + ["rand_like"] * len(cols["K"]), kinds
            noise = log.draws[0][1] if i > 0 else torch.zeros_like(z)
            uniforms = torch.cat([d[1] for d in log.draws[(1 if i > 0 else 0):]], dim=1)
            mean, _, logvar = diff.gauss_q_posterior(x0, z, t)
            idx_new = ref_mixed.ohe_to_categories(torch.exp(log_z_new).round(), diff.categorical_num_classes)
            p = f"t{i}."
            arrays.update({
                p + "model_out": np32(model_out), p + "z": np32(z), p + "idx": np32(idx).astype(np.uint8),
                p + "noise": np32(noise), p + "uniforms": np32(uniforms), p + "z_new": np32(z_new),
                p + "x0": np32(x0), p + "mean": np32(mean), p + "log_theta": np32(log_theta),
                p + "idx_new": np32(idx_new).astype(np.uint8)})
        save(f"reverse_steps_{colname}_T1000", **arrays)


def gen_sample_chain():
    """Full (short) reverse chains through the reference's own sample()."""
    for name, colname, model, bidir, T in (("eicu_lstm_T8", "eicu", "lstm", True, 8),
                                           ("k3mix_gru_T6", "k3mix", "gru", True, 6),
                                           ("hirid_lstm_uni_T5", "hirid", "lstm", False, 5)):
        cols = COLUMNS[colname]
        L, H, N, B = 16, 24, 2, 3
        rnn, diff = build(cols, L, H, N, T, model=model, bidir=bidir)
        n_num, n_cat, sumK = len(cols["num"]), len(cols["cat"]), sum(cols["K"])
        outs = []
        orig = rnn.forward

        def hooked(x, time, cond=None):
            o = orig(x, time, cond)
            outs.append((x.detach().clone(), o.detach().clone()))
            return o

        rnn.forward = hooked
        torch.manual_seed(21)
        with DrawLog() as log:
            sample = diff.sample(B)
        rnn.forward = orig
        draws = list(log.draws)
        assert draws[0][0] == "randn"
        z_T = draws[0][1]
        u_prior = torch.cat([d[1] for d in draws[1:1 + n_cat]], dim=1)
        pos = 1 + n_cat
        noises, unis = [], []
        for i in reversed(range(T)):
            if i > 0:
                assert draws[pos][0] == "randn_like"
                noises.append(draws[pos][1]); pos += 1
            else:
                noises.append(torch.zeros_like(z_T))
            unis.append(torch.cat([d[1] for d in draws[pos:pos + n_cat]], dim=1)); pos += n_cat
        assert pos == len(draws)
        arrays = state_arrays(diff)
        arrays.update(
            z_T=np32(z_T), u_prior=np32(u_prior), noise=np32(torch.stack(noises)), uniforms=np32(torch.stack(unis)),
            x_in=np32(torch.stack([o[0] for o in outs])), model_out=np32(torch.stack([o[1] for o in outs])),
            sample=np32(sample), num_idx=np.array(cols["num"]), cat_idx=np.array(cols["cat"]), K=np.array(cols["K"]),
            cfg=np.array([L, H, N, T, int(bidir), int(model == "gru")]))
        save(f"sample_chain_{name}", **arrays)


def gen_gaussian():
    torch.manual_seed(3)
    C, L, H, N, T, B = 6, 12, 16, 2, 6, 4
    rnn = ref_blocks.RNN(C, H, C, N, model="lstm", bidirectional=True)
    diff = ref_gauss.GaussianDiffusion(rnn, seq_length=L, channels=C, timesteps=T)
    data = torch.rand(B, C, L)
    t = torch.tensor([0, 1, 3, 5])
    with DrawLog() as log:
        loss = diff.p_losses(diff.normalize(data.clone()), t)
    loss.backward()
    noise = log.draws[0][1]
    arrays = state_arrays(diff)
    arrays.update({"grad." + k: np32(p.grad) for k, p in diff.named_parameters()})
    arrays.update(data=np32(data), t=np32(t), noise=np32(noise), loss=np32(loss))
    torch.manual_seed(4)
    with DrawLog() as log:
        sample = diff.sample(3)
    draws = log.draws
    arrays.update(z_T=np32(draws[0][1]),
                  chain_noise=np32(torch.stack([d[1] for d in draws[1:]] + [torch.zeros_like(draws[0][1])])),
                  sample=np32(sample), cfg=np.array([C, L, H, N, T]))
    save("gaussian_T6", **arrays)


def gen_train_steps():
    """Two optimizer steps of the reference loop (et_diff.py:226-261) in plain torch:
    loss / ga -> backward (x ga) -> clip_grad_norm_(1.0) -> Adam(lr, betas=(0.9, 0.99)) -> zero_grad."""
    cols = COLUMNS["eicu"]
    L, H, N, T, B, ga = 12, 16, 2, 50, 4, 2
    rnn, diff = build(cols, L, H, N, T)
    arrays = {"init." + k: np32(v) for k, v in diff.state_dict().items()}
    opt = torch.optim.Adam(diff.parameters(), lr=1e-3, betas=(0.9, 0.99))
    gen = torch.Generator().manual_seed(17)
    torch.manual_seed(23)
    batches, ts, noises, unis, losses, norms = [], [], [], [], [], []
    for step in range(2):
        total = 0.0
        for _ in range(ga):
            data = make_data(B, cols, L, gen)
            batches.append(data.clone())
            with DrawLog() as log:
                loss = diff(data) / ga
            loss.backward()
            total += loss.item()
            ts.append(log.draws[0][1]); noises.append(log.draws[1][1])
            unis.append(torch.cat([d[1] for d in log.draws[2:]], dim=1))
        norms.append(torch.nn.utils.clip_grad_norm_(diff.parameters(), 1.0))
        opt.step(); opt.zero_grad()
        losses.append(total)
        arrays.update({f"step{step}." + k: np32(v) for k, v in diff.state_dict().items() if k.startswith("model.")})
    arrays.update(data=np32(torch.stack(batches)), t=np32(torch.stack(ts)), noise=np32(torch.stack(noises)),
                  uniforms=np32(torch.stack(unis)), total_loss=np.array(losses, dtype=np.float32),
                  grad_norm=np32(torch.stack(norms)), cfg=np.array([L, H, N, T, B, ga]), lr=np.float32(1e-3),
                  num_idx=np.array(cols["num"]), cat_idx=np.array(cols["cat"]), K=np.array(cols["K"]))
    save("train_steps_eicu", **arrays)


def gen_sliced_lse():
    gen = torch.Generator().manual_seed(1)
    x = torch.randn(5, 9, 7, generator=gen) * 3 - 2
    offsets = torch.tensor([0, 3, 5, 9])
    save("sliced_logsumexp", x=np32(x), offsets=np32(offsets), out=np32(ref_mixed.sliced_logsumexp(x, offsets)))


CLASSIFIER_CASES = [("lstm", 9, 64, 50, 24), ("gru", 5, 64, 33, 17), ("lstm", 23, 40, 70, 30), ("gru", 9, 256, 40, 12)]


def gen_classifier():
    """the TSTR classifier of the reference itself (evaluate/nn_classifier.py:89-127, RNN): forward
    probabilities, binary cross entropy (what nnTimeSeriesClassifier.train minimises, :141-142, :158-159) and the gradient
    of every parameter, for the GPU test to compare td_cls_forward / td_cls_backward with."""
    import evaluate.nn_classifier as ref_cls
    assert os.path.realpath(ref_cls.__file__).startswith(os.path.realpath(REF) + os.sep), ref_cls.__file__
    for model, C, H, B, L in CLASSIFIER_CASES:
        torch.manual_seed(C + H)
        net = ref_cls.RNN(in_channels=C, hidden_dim=H, model=model)
        X = torch.randn(B, C, L)
        y = (torch.rand(B) > 0.5).float()
        prob = net(X)
        loss = torch.nn.functional.binary_cross_entropy(prob, y)
        loss.backward()
        arrays = {"sd." + k: np32(v) for k, v in net.state_dict().items()}
        arrays.update({"grad." + k: np32(v.grad) for k, v in net.named_parameters()})
        save(f"classifier_{model}_c{C}_h{H}", X=np32(X), y=np32(y), prob=np32(prob), loss=np32(loss), cfg=np.array([C, H, B, L]), **arrays)


if __name__ == "__main__":
    torch.set_num_threads(4)
    gen_schedules()
    gen_sliced_lse()
    gen_rnn_forward()
    gen_mixed_loss()
    gen_reverse_steps()
    gen_sample_chain()
    gen_gaussian()
    gen_train_steps()
    gen_classifier()

"""Pins the CPU oracle (oracle/) to the golden vectors captured from the reference
(tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest

from conftest import load_golden, sub
from oracle import oracle as orc

SCHEDULE_KEYS = [
    "betas", "alphas_cumprod", "alphas_cumprod_prev", "log_alpha", "log_1_min_alpha", "log_cumprod_alpha",
    "log_1_min_cumprod_alpha", "sqrt_alphas_cumprod", "sqrt_one_minus_alphas_cumprod",
    "log_one_minus_alphas_cumprod", "sqrt_recip_alphas_cumprod", "sqrt_recipm1_alphas_cumprod",
    "posterior_variance", "posterior_log_variance_clipped", "posterior_mean_coef1", "posterior_mean_coef2",
    "loss_weight"]


def tables_of(g, prefix="sd."):
    return {k: g[prefix + k] for k in SCHEDULE_KEYS}


@pytest.mark.parametrize("kind,T", [("cosine", 1000), ("cosine", 8), ("linear", 50), ("linear", 1000)])
def test_schedule(kind, T):
    g = load_golden(f"schedule_{kind}_T{T}")
    s = orc.schedule(T, kind)
    for k in SCHEDULE_KEYS:
        # float64 -> float32 of the same formulas: allow 1 ulp for libm differences (numpy vs torch)
        np.testing.assert_allclose(s[k], g[k], rtol=2e-7, atol=1e-30, err_msg=k)


def test_sliced_logsumexp():
    g = load_golden("sliced_logsumexp")
    out = orc.sliced_logsumexp(g["x"], g["offsets"])
    # the log_sub_exp cancellation amplifies 1-ulp libm differences
    np.testing.assert_allclose(out, g["out"], rtol=0, atol=2e-5)


@pytest.mark.parametrize("name", ["lstm_bi", "lstm_uni", "gru_bi", "gru_uni", "lstm_bi_h92"])
def test_rnn_forward(name):
    g = load_golden("rnn_" + name)
    den = orc.Denoiser(sub(g, "sd."))
    temb, _ = den.time_embed(g["time"])
    np.testing.assert_allclose(temb, g["temb"], rtol=1e-4, atol=2e-5)
    y, hidden = den.forward(g["x"], g["time"], want_hidden=True)
    np.testing.assert_allclose(hidden, g["hidden"], rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(y, g["y"], rtol=1e-4, atol=5e-5)


@pytest.mark.parametrize("name", ["lstm_bi", "lstm_uni", "gru_bi", "gru_uni"])
def test_rnn_backward(name):
    g = load_golden("rnn_" + name)
    den = orc.Denoiser(sub(g, "sd."))
    y, gx, grads = den.backward(g["x"], g["time"], g["grad_y"])
    np.testing.assert_allclose(y, g["y"], rtol=1e-4, atol=5e-5)
    np.testing.assert_allclose(gx, g["grad_x"], rtol=1e-3, atol=2e-5)
    for k, v in grads.items():
        ref = g["grad." + k]
        scale = max(1.0, float(np.abs(ref).max()))
        np.testing.assert_allclose(v, ref, rtol=1e-3, atol=1e-4 * scale, err_msg=k)


def test_rnn_bf16_emulation_close_to_fp32():
    """The bf16-operand emulation (what the MFMA kernel computes) stays within the stated bf16
    tolerance of the fp32 denoiser: 3e-2 absolute on outputs of O(1)."""
    g = load_golden("rnn_lstm_bi")
    y32 = orc.Denoiser(sub(g, "sd.")).forward(g["x"], g["time"])
    y16 = orc.Denoiser(sub(g, "sd."), bf16=True).forward(g["x"], g["time"])
    assert np.abs(y16 - y32).max() < 3e-2
    assert np.abs(y16 - y32).max() > 0  # the switch does something


def _diff_from(g, L, T, prefix="sd."):
    return orc.Diffusion(len(g["num_idx"]), g["K"], L, T, tables=tables_of(g, prefix))


def _split(g, data):
    x_num = data[:, g["num_idx"], :] * 2 - 1          # normalize_numericals, mixed_diffusion.py:189-194
    x_idx = data[:, g["cat_idx"], :].astype(np.uint8)
    return x_num, x_idx


@pytest.mark.parametrize("name", ["eicu_T50", "k3mix_T1000", "hirid_lin_T50"])
def test_q_sample_and_loss(name):
    g = load_golden("mixed_loss_" + name)
    L, H, N, T = [int(v) for v in g["cfg"]]
    d = _diff_from(g, L, T)
    x_num, x_idx = _split(g, g["data"])
    x_in, xt_idx = d.q_sample(x_num, x_idx, g["t"], g["noise"], g["uniforms"])
    np.testing.assert_allclose(x_in, g["x_in"], rtol=1e-6, atol=1e-6)

    den = orc.Denoiser(sub(g, "sd.model."))
    mo = den.forward(x_in, g["t"])
    np.testing.assert_allclose(mo, g["model_out"], rtol=1e-4, atol=5e-5)

    lam = float(g["loss_lambda"])
    # reference-faithful normalisation (cumulative logcumsumexp differences)
    r = d.mixed_loss(g["model_out"], g["noise"], x_idx, xt_idx, g["t"], lam, faithful=True)
    np.testing.assert_allclose(r["log_true"], g["log_true"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(r["log_model"], g["log_model"], rtol=0, atol=3e-5)
    np.testing.assert_allclose(r["Lt"], g["Lt"], rtol=2e-4, atol=2e-4)
    np.testing.assert_allclose(r["kl_prior"], g["kl_prior"], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(r["loss_gauss"], g["loss_gauss"], rtol=1e-5)
    np.testing.assert_allclose(r["loss"], g["loss"], rtol=2e-4)
    # direct per-slice logsumexp (what the HIP kernel does): same loss within the documented slack
    r2 = d.mixed_loss(g["model_out"], g["noise"], x_idx, xt_idx, g["t"], lam, faithful=False)
    np.testing.assert_allclose(r2["loss"], g["loss"], rtol=2e-3)
    gscale = float(np.abs(g["grad_model_out"]).max())
    # the reference's autograd differentiates through its noisy cumulative normalisation at t <= ~5
    np.testing.assert_allclose(r2["grad"], g["grad_model_out"], rtol=2e-3, atol=5e-3 * gscale)


@pytest.mark.parametrize("name", ["eicu", "k3mix"])
def test_reverse_steps(name):
    g = load_golden(f"reverse_steps_{name}_T1000")
    s = orc.schedule(1000, "cosine")
    d = orc.Diffusion(len(g["num_idx"]), g["K"], 12, 1000, tables=s)
    for i in g["steps"]:
        p = f"t{i}."
        for faithful in (True, False):
            z_new, idx_new, det = d.reverse_step(g[p + "model_out"], g[p + "z"], g[p + "idx"], int(i), g[p + "noise"],
                                                 g[p + "uniforms"], faithful=faithful, details=True)
            np.testing.assert_allclose(det["x0"], g[p + "x0"], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(det["mean"], g[p + "mean"], rtol=1e-5, atol=1e-5)
            np.testing.assert_allclose(z_new, g[p + "z_new"], rtol=1e-5, atol=1e-5)
            # log theta: faithful matches tightly; the direct form differs by a per-slice shift that
            # reaches ~2e-2 at small t -- the argmax is invariant to it.
            # (the cancellation is chaotic at the 1e-2 level for t <= 10, so even the faithful form
            # only tracks the fixture there.)
            tol = 3e-2 if i <= 10 else 1e-5
            np.testing.assert_allclose(det["log_theta"], g[p + "log_theta"], rtol=0, atol=tol)
            margin = d.gumbel_margin(g[p + "log_theta"], g[p + "uniforms"])
            safe = margin > (1e-3 if i <= 10 else 1e-5)
            assert safe.mean() > 0.99
            assert np.array_equal(idx_new[safe], g[p + "idx_new"][safe])


@pytest.mark.parametrize("name", ["eicu_lstm_T8", "k3mix_gru_T6", "hirid_lstm_uni_T5"])
def test_sample_chain(name):
    g = load_golden("sample_chain_" + name)
    L, H, N, T, bidir, gru = [int(v) for v in g["cfg"]]
    d = _diff_from(g, L, T)
    den = orc.Denoiser(sub(g, "sd.model."))
    idx_T = d.prior_categorical(g["u_prior"])
    z, idx, rec = d.sample_chain(den, g["z_T"], idx_T, g["noise"], g["uniforms"], record=True)
    np.testing.assert_allclose(rec[0][0], g["x_in"][0], rtol=1e-6, atol=1e-6)
    for s in range(T):
        np.testing.assert_allclose(rec[s][0], g["x_in"][s], rtol=1e-3, atol=2e-4, err_msg=f"x_in step {s}")
        np.testing.assert_allclose(rec[s][1], g["model_out"][s], rtol=1e-3, atol=5e-4, err_msg=f"model_out step {s}")
    out = d.decode(z, idx, g["num_idx"], g["cat_idx"], channels=d.n_num + d.sumK)
    np.testing.assert_allclose(out, g["sample"], rtol=1e-3, atol=5e-4)


def test_gaussian():
    g = load_golden("gaussian_T6")
    C_, L, H, N, T = [int(v) for v in g["cfg"]]
    d = orc.Diffusion(C_, [], L, T, kind="cosine")
    den = orc.Denoiser(sub(g, "sd.model."))
    x0 = g["data"] * 2 - 1
    empty_idx = np.zeros((x0.shape[0], 0, L), np.uint8)
    x_in, _ = d.q_sample(x0, empty_idx, g["t"], g["noise"], np.zeros((x0.shape[0], 0, L), np.float32))
    mo = den.forward(x_in, g["t"])
    r = d.mixed_loss(mo, g["noise"], empty_idx, empty_idx, g["t"], 0.0)
    np.testing.assert_allclose(r["loss"], g["loss"], rtol=2e-4)
    B = g["z_T"].shape[0]
    z, _ = d.sample_chain(den, g["z_T"], np.zeros((B, 0, L), np.uint8), g["chain_noise"], None)
    np.testing.assert_allclose((z + 1) * 0.5, g["sample"], rtol=1e-3, atol=5e-4)


def test_train_steps():
    """Two optimizer steps of the reference loop (et_diff.py:226-261): loss/ga, backward x ga,
    clip_grad_norm_(1.0), Adam(betas=(0.9, 0.99))."""
    g = load_golden("train_steps_eicu")
    L, H, N, T, B, ga = [int(v) for v in g["cfg"]]
    d = orc.Diffusion(len(g["num_idx"]), g["K"], L, T, tables=tables_of(g, "init."))
    params = {k: v.copy() for k, v in sub(g, "init.model.").items()}
    m = {k: np.zeros_like(v) for k, v in params.items()}
    v = {k: np.zeros_like(p) for k, p in params.items()}
    mb = 0
    for step in range(2):
        grads = {k: np.zeros_like(p) for k, p in params.items()}
        total = 0.0
        for _ in range(ga):
            den = orc.Denoiser(params)
            x_num, x_idx = _split(g, g["data"][mb])
            t = g["t"][mb]
            x_in, xt_idx = d.q_sample(x_num, x_idx, t, g["noise"][mb], g["uniforms"][mb])
            mo = den.forward(x_in, t)
            r = d.mixed_loss(mo, g["noise"][mb], x_idx, xt_idx, t, 0.8, faithful=True)
            total += r["loss"] / ga
            _, _, gp = den.backward(x_in, t, r["grad"] / ga)
            for k in grads:
                grads[k] += gp[k]
            mb += 1
        norm = orc.clip_and_adam(params, grads, m, v, step + 1, float(g["lr"]))
        np.testing.assert_allclose(total, g["total_loss"][step], rtol=5e-4)
        np.testing.assert_allclose(norm, g["grad_norm"][step], rtol=5e-3)
        for k in params:
            # Adam's first steps move every weight by ~lr * sign(g): where |g| ~ eps a tiny gradient
            # difference changes the step, so bound the error by 2 lr and its frequency by 1 %.
            err = np.abs(params[k] - g[f"step{step}.model." + k])
            assert err.max() < 2e-3, k
            assert (err > 1e-4).mean() < 0.01, k

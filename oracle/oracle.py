"""ctypes front end of the CPU oracle (oracle/etdiff_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package timediff_amd never does.  Parity pin: tests/test_oracle_golden.py checks every function here
against tests/golden/*.npz, which were produced by running the reference's Python modules
(tests/golden/make_golden.py).
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libetdiff_oracle.so")
_lib = None


def build(force=False):
    src = os.path.join(_HERE, "etdiff_oracle.c")
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libetdiff_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _ptr(a, ty=C.c_float):
    return None if a is None else a.ctypes.data_as(C.POINTER(ty))


# ---------------------------------------------------------------------------------------------
# Schedules: utils.py:75-87 and mixed_diffusion.py:103-158 restated in numpy float64 -> float32
# ---------------------------------------------------------------------------------------------
def schedule(T, kind="cosine"):
    if kind == "linear":
        scale = 1000 / T
        betas = np.linspace(scale * 0.0001, scale * 0.02, T, dtype=np.float64)
    elif kind == "cosine":
        s = 0.008
        x = np.linspace(0, T, T + 1, dtype=np.float64)
        ac = np.cos(((x / T) + s) / (1 + s) * math.pi * 0.5) ** 2
        ac = ac / ac[0]
        betas = np.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
    else:
        raise ValueError(f"unknown beta schedule {kind}")
    alphas = 1.0 - betas
    ac = np.cumprod(alphas)
    ac_prev = np.concatenate([[1.0], ac[:-1]])
    log_alpha = np.log(alphas)
    log_cum = np.cumsum(log_alpha)
    with np.errstate(divide="ignore"):
        out = dict(
            betas=betas, alphas_cumprod=ac, alphas_cumprod_prev=ac_prev,
            log_alpha=log_alpha, log_1_min_alpha=np.log(1 - np.exp(log_alpha) + 1e-40),
            log_cumprod_alpha=log_cum, log_1_min_cumprod_alpha=np.log(1 - np.exp(log_cum) + 1e-40),
            sqrt_alphas_cumprod=np.sqrt(ac), sqrt_one_minus_alphas_cumprod=np.sqrt(1 - ac),
            log_one_minus_alphas_cumprod=np.log(1 - ac), sqrt_recip_alphas_cumprod=np.sqrt(1 / ac),
            sqrt_recipm1_alphas_cumprod=np.sqrt(1 / ac - 1),
        )
        pv = betas * (1 - ac_prev) / (1 - ac)
        out.update(
            posterior_variance=pv, posterior_log_variance_clipped=np.log(np.maximum(pv, 1e-20)),
            posterior_mean_coef1=betas * np.sqrt(ac_prev) / (1 - ac),
            posterior_mean_coef2=(1 - ac_prev) * np.sqrt(alphas) / (1 - ac),
            loss_weight=np.ones_like(ac),
        )
    return {k: v.astype(np.float32) for k, v in out.items()}


# ---------------------------------------------------------------------------------------------
# Denoiser
# ---------------------------------------------------------------------------------------------
class RnnCfg(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("C_in", "H", "C_out", "N", "D", "gru", "embed_dim", "time_dim", "bf16")]


MISC_KEYS = ["time_emb.1.weight", "time_emb.1.bias", "time_emb.3.weight", "time_emb.3.bias",
             "rnn.time_mlp.1.weight", "rnn.time_mlp.1.bias", "rnn.norm.weight", "rnn.norm.bias",
             "fc.weight", "fc.bias"]


def layer_keys(N, D):
    keys = []
    for l in range(N):
        for d in range(D):
            sfx = f"_l{l}" + ("_reverse" if d else "")
            keys += [f"rnn.rnn.{n}{sfx}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    return keys


class Denoiser:
    """RNN (blocks.py:300-341) from a {name: ndarray} dict keyed like RNN.state_dict()."""

    def __init__(self, sd, prefix="", bf16=False):
        sd = {k[len(prefix):]: _f32(v) for k, v in sd.items() if k.startswith(prefix)}
        w_ih0 = sd["rnn.rnn.weight_ih_l0"]
        w_hh0 = sd["rnn.rnn.weight_hh_l0"]
        H = w_hh0.shape[1]
        G = w_hh0.shape[0] // H
        D = 2 if "rnn.rnn.weight_ih_l0_reverse" in sd else 1
        N = 0
        while f"rnn.rnn.weight_ih_l{N}" in sd:
            N += 1
        self.cfg = RnnCfg(C_in=w_ih0.shape[1], H=H, C_out=sd["fc.weight"].shape[0], N=N, D=D, gru=int(G == 3),
                          embed_dim=sd["time_emb.1.weight"].shape[1], time_dim=sd["time_emb.1.weight"].shape[0],
                          bf16=int(bf16))
        self.lkeys = layer_keys(N, D)
        self.lw = [sd[k] for k in self.lkeys]
        self.misc = [sd[k] for k in MISC_KEYS]
        self._lw = (C.POINTER(C.c_float) * len(self.lw))(*[_ptr(a) for a in self.lw])
        self._misc = (C.POINTER(C.c_float) * len(self.misc))(*[_ptr(a) for a in self.misc])

    def time_embed(self, t):
        t = np.ascontiguousarray(t, dtype=np.int64)
        B = t.shape[0]
        temb = np.empty((B, self.cfg.time_dim), np.float32)
        film = np.empty((B, 2 * self.cfg.D * self.cfg.H), np.float32)
        lib().or_time_embed(C.byref(self.cfg), self._misc, _ptr(t, C.c_int64), B, _ptr(temb), _ptr(film))
        return temb, film

    def _set_masks(self, masks, B, L):
        """masks: (N-1, B, L, D*H) multiplicative inter-layer dropout masks (entries 0 or 1/(1-p)), or None."""
        if masks is None:
            lib().or_set_dropout_masks(None, 0)
            return None
        masks = _f32(masks)
        assert masks.shape == (self.cfg.N - 1, B, L, self.cfg.D * self.cfg.H), masks.shape
        lib().or_set_dropout_masks(_ptr(masks), B)
        return masks

    def forward(self, x, t, want_hidden=False, masks=None):
        x = _f32(x)
        t = np.ascontiguousarray(t, dtype=np.int64)
        B, _, L = x.shape
        y = np.empty((B, self.cfg.C_out, L), np.float32)
        hidden = np.empty((B, L, self.cfg.D * self.cfg.H), np.float32) if want_hidden else None
        masks = self._set_masks(masks, B, L)
        lib().or_rnn_forward(C.byref(self.cfg), self._lw, self._misc, _ptr(x), _ptr(t, C.c_int64), B, L, _ptr(y),
                             _ptr(hidden))
        self._set_masks(None, B, L)
        return (y, hidden) if want_hidden else y

    def backward(self, x, t, grad_y, masks=None):
        """-> (y, grad_x, {param name: grad})"""
        x = _f32(x)
        gy = _f32(grad_y)
        t = np.ascontiguousarray(t, dtype=np.int64)
        B, _, L = x.shape
        glw = [np.zeros_like(a) for a in self.lw]
        gmisc = [np.zeros_like(a) for a in self.misc]
        _glw = (C.POINTER(C.c_float) * len(glw))(*[_ptr(a) for a in glw])
        _gmisc = (C.POINTER(C.c_float) * len(gmisc))(*[_ptr(a) for a in gmisc])
        gx = np.empty_like(x)
        y = np.empty((B, self.cfg.C_out, L), np.float32)
        masks = self._set_masks(masks, B, L)
        lib().or_rnn_backward(C.byref(self.cfg), self._lw, self._misc, _ptr(x), _ptr(t, C.c_int64), B, L, _ptr(gy),
                              _glw, _gmisc, _ptr(gx), _ptr(y))
        self._set_masks(None, B, L)
        grads = dict(zip(self.lkeys, glw))
        grads.update(dict(zip(MISC_KEYS, gmisc)))
        return y, gx, grads


# ---------------------------------------------------------------------------------------------
# Diffusion math
# ---------------------------------------------------------------------------------------------
class _Diff(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("n_num", "n_cat", "sumK", "L", "T")] + [("K", C.POINTER(C.c_int32))] + [
        (n, C.POINTER(C.c_float)) for n in ("sqrt_ac", "sqrt_1m_ac", "sqrt_recip_ac", "sqrt_recipm1_ac", "post_c1",
                                            "post_c2", "post_logvar", "log_alpha", "log_1m_alpha", "log_cum_alpha",
                                            "log_1m_cum_alpha")]


_TABLE_MAP = dict(sqrt_ac="sqrt_alphas_cumprod", sqrt_1m_ac="sqrt_one_minus_alphas_cumprod",
                  sqrt_recip_ac="sqrt_recip_alphas_cumprod", sqrt_recipm1_ac="sqrt_recipm1_alphas_cumprod",
                  post_c1="posterior_mean_coef1", post_c2="posterior_mean_coef2",
                  post_logvar="posterior_log_variance_clipped", log_alpha="log_alpha", log_1m_alpha="log_1_min_alpha",
                  log_cum_alpha="log_cumprod_alpha", log_1m_cum_alpha="log_1_min_cumprod_alpha")


class Diffusion:
    """The MixedDiffusion arithmetic (mixed_diffusion.py) on numpy arrays.  K = [] gives the
    Gaussian-only process of gaussian_diffusion.py."""

    def __init__(self, n_num, K, L, T, tables=None, kind="cosine"):
        self.n_num, self.K, self.L, self.T = n_num, [int(k) for k in K], L, T
        self.n_cat, self.sumK = len(self.K), int(sum(self.K))
        self.tables = {k: _f32(v) for k, v in (tables or schedule(T, kind)).items()}
        self._K = np.asarray(self.K if self.K else [0], dtype=np.int32)
        self.d = _Diff(n_num=n_num, n_cat=self.n_cat, sumK=self.sumK, L=L, T=T, K=_ptr(self._K, C.c_int32),
                       **{f: _ptr(self.tables[name]) for f, name in _TABLE_MAP.items()})

    def _u8(self, a):
        return np.ascontiguousarray(a, dtype=np.uint8)

    def q_sample(self, x0_num, x0_idx, t, noise, uniforms):
        B = x0_num.shape[0]
        x0_num, noise, uniforms = _f32(x0_num), _f32(noise), _f32(uniforms)
        x0_idx = self._u8(x0_idx)
        t = np.ascontiguousarray(t, dtype=np.int64)
        x_in = np.empty((B, self.n_num + self.sumK, self.L), np.float32)
        xt_idx = np.empty((B, self.n_cat, self.L), np.uint8)
        lib().or_q_sample_mixed(C.byref(self.d), B, _ptr(x0_num), _ptr(x0_idx, C.c_uint8), _ptr(t, C.c_int64),
                                _ptr(noise), _ptr(uniforms), _ptr(x_in), _ptr(xt_idx, C.c_uint8))
        return x_in, xt_idx

    def reverse_step(self, model_out, z, idx, t, noise, uniforms, faithful=False, details=False):
        B = model_out.shape[0]
        model_out, z, noise, uniforms = _f32(model_out), _f32(z), _f32(noise), _f32(uniforms)
        idx = self._u8(idx)
        t = np.ascontiguousarray(np.broadcast_to(t, (B,)), dtype=np.int64)
        z_new = np.empty_like(z)
        idx_new = np.empty_like(idx)
        x0 = np.empty_like(z) if details else None
        mean = np.empty_like(z) if details else None
        lth = np.empty((B, self.sumK, self.L), np.float32) if details else None
        lib().or_reverse_step_mixed(C.byref(self.d), B, _ptr(model_out), _ptr(z), _ptr(idx, C.c_uint8),
                                    _ptr(t, C.c_int64), _ptr(noise), _ptr(uniforms), int(faithful), _ptr(z_new),
                                    _ptr(idx_new, C.c_uint8), _ptr(x0), _ptr(mean), _ptr(lth))
        if details:
            return z_new, idx_new, dict(x0=x0, mean=mean, log_theta=lth)
        return z_new, idx_new

    def gumbel_margin(self, log_theta, uniforms):
        log_theta, uniforms = _f32(log_theta), _f32(uniforms)
        B = log_theta.shape[0]
        m = np.empty((B, self.n_cat, self.L), np.float32)
        lib().or_gumbel_margin(C.byref(self.d), B, _ptr(log_theta), _ptr(uniforms), _ptr(m))
        return m

    def mixed_loss(self, model_out, noise, x0_idx, xt_idx, t, loss_lambda, faithful=False):
        B = model_out.shape[0]
        model_out, noise = _f32(model_out), _f32(noise)
        x0_idx, xt_idx = self._u8(x0_idx), self._u8(xt_idx)
        t = np.ascontiguousarray(t, dtype=np.int64)
        loss = np.zeros(1, np.float32)
        lg = np.zeros(1, np.float32)
        grad = np.zeros_like(model_out)
        Lt = np.zeros(B, np.float32)
        klp = np.zeros(B, np.float32)
        ltrue = np.zeros((B, self.sumK, self.L), np.float32)
        lmodel = np.zeros((B, self.sumK, self.L), np.float32)
        lib().or_mixed_loss(C.byref(self.d), B, _ptr(model_out), _ptr(noise), _ptr(x0_idx, C.c_uint8),
                            _ptr(xt_idx, C.c_uint8), _ptr(t, C.c_int64), C.c_float(loss_lambda), int(faithful),
                            _ptr(loss), _ptr(grad), _ptr(Lt), _ptr(klp), _ptr(ltrue), _ptr(lmodel), _ptr(lg))
        return dict(loss=float(loss[0]), grad=grad, Lt=Lt, kl_prior=klp, log_true=ltrue, log_model=lmodel,
                    loss_gauss=float(lg[0]))

    def prior_categorical(self, uniforms):
        uniforms = _f32(uniforms)
        B = uniforms.shape[0]
        idx = np.empty((B, self.n_cat, self.L), np.uint8)
        lib().or_prior_categorical(C.byref(self.d), B, _ptr(uniforms), _ptr(idx, C.c_uint8))
        return idx

    def build_x_in(self, z, idx):
        z, idx = _f32(z), self._u8(idx)
        B = z.shape[0]
        x_in = np.empty((B, self.n_num + self.sumK, self.L), np.float32)
        lib().or_build_x_in(C.byref(self.d), B, _ptr(z), _ptr(idx, C.c_uint8), _ptr(x_in))
        return x_in

    def sample_chain(self, denoiser, z_T, idx_T, noise, uniforms, record=False):
        """mixed_diffusion.py:437-446 with external draws: noise/uniforms are (T, ...) in loop order
        (index 0 is step t = T-1).  -> final (z, idx) before decode."""
        z, idx = _f32(z_T).copy(), self._u8(idx_T).copy()
        B = z.shape[0]
        rec = []
        for s, i in enumerate(reversed(range(self.T))):
            t = np.full((B,), i, np.int64)
            x_in = self.build_x_in(z, idx) if self.n_cat else z
            mo = denoiser.forward(x_in, t)
            if record:
                rec.append((x_in, mo))
            u = uniforms[s] if self.n_cat else np.zeros((B, 0, self.L), np.float32)
            z, idx = self.reverse_step(mo, z, idx, t, noise[s], u)
        return (z, idx, rec) if record else (z, idx)

    def decode(self, z, idx, num_idx, cat_idx, channels=None):
        """mixed_diffusion.py:448-453, :173-179: (z+1)/2 and class indices scattered to dataset channel
        order.  The reference sizes the result `channels - n_cat` (channels = denoiser width), which is
        n_num + n_cat only when every variable is binary; extra channels stay zero."""
        B = z.shape[0]
        C_out = len(num_idx) + len(cat_idx) if channels is None else channels - len(cat_idx)
        out = np.zeros((B, C_out, self.L), np.float32)
        out[:, num_idx, :] = (z + 1) * 0.5
        if len(cat_idx):
            out[:, cat_idx, :] = idx.astype(np.float32)
        return out


def sliced_logsumexp(x, offsets):
    x = _f32(x)
    offsets = np.ascontiguousarray(offsets, dtype=np.int64)
    out = np.empty_like(x)
    B, S, L = x.shape
    lib().or_sliced_logsumexp(_ptr(x), B, S, L, _ptr(offsets, C.c_int64), len(offsets) - 1, _ptr(out))
    return out


def philox_normals(seed, patient, l, tag, n):
    out = np.empty(n, np.float32)
    lib().or_philox_normals(C.c_uint64(seed), C.c_uint32(patient), C.c_uint32(l), C.c_uint32(tag), n, _ptr(out))
    return out


def philox_uniforms(seed, patient, l, tag, n):
    out = np.empty(n, np.float32)
    lib().or_philox_uniforms(C.c_uint64(seed), C.c_uint32(patient), C.c_uint32(l), C.c_uint32(tag), n, _ptr(out))
    return out


# ---------------------------------------------------------------------------------------------
# Trainer arithmetic (et_diff.py:239, :75, :247): clip_grad_norm_(1.0) then Adam, in numpy fp32.
# ---------------------------------------------------------------------------------------------
def clip_and_adam(params, grads, m, v, step, lr, betas=(0.9, 0.99), eps=1e-8, max_norm=1.0):
    """In place on dicts of fp32 arrays; step counts from 1.  -> total grad norm before clipping."""
    total = math.sqrt(sum(float(np.sum(g.astype(np.float64) ** 2)) for g in grads.values()))
    coef = min(1.0, max_norm / (total + 1e-6))
    b1, b2 = betas
    for k in params:
        g = grads[k] * np.float32(coef)
        m[k] = b1 * m[k] + (1 - b1) * g
        v[k] = b2 * v[k] + (1 - b2) * g * g
        step_size = lr / (1 - b1 ** step)
        denom = np.sqrt(v[k]) / math.sqrt(1 - b2 ** step) + eps
        params[k] -= (step_size * m[k] / denom).astype(np.float32)
    return total

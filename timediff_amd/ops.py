"""Tensor-level front end of the HIP library: torch tensors in, torch tensors out.

torch is used for device memory and stream handles only; every computation below runs in
libtimediff_hip.so.  All tensors must live on the current ROCm device.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import DiffusionConfig, Rng, RnnConfig, check, lib, ptr, ptr_table


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t):
    assert t.is_cuda, "tensor must be on the GPU"
    return t.detach().to(torch.float32).contiguous()


def layer_keys(layers, dirs):
    keys = []
    for l in range(layers):
        for d in range(dirs):
            sfx = f"_l{l}" + ("_reverse" if d else "")
            keys += [f"rnn.rnn.{n}{sfx}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    return keys


class DenoiserKernels:
    """The denoiser (blocks.py RNN) on the HIP kernels, from a state_dict-like mapping of CUDA tensors."""

    def __init__(self, state, prefix=""):
        sd = {k[len(prefix):]: v for k, v in state.items() if k.startswith(prefix)}
        w_ih0, w_hh0 = sd["rnn.rnn.weight_ih_l0"], sd["rnn.rnn.weight_hh_l0"]
        hidden = w_hh0.shape[1]
        gates = w_hh0.shape[0] // hidden
        dirs = 2 if "rnn.rnn.weight_ih_l0_reverse" in sd else 1
        layers = 0
        while f"rnn.rnn.weight_ih_l{layers}" in sd:
            layers += 1
        self.cfg = RnnConfig(c_in=w_ih0.shape[1], hidden=hidden, c_out=sd["fc.weight"].shape[0], layers=layers,
                             dirs=dirs, gru=int(gates == 3), embed_dim=sd["time_emb.1.weight"].shape[1],
                             time_dim=sd["time_emb.1.weight"].shape[0])
        self.device = w_ih0.device
        self._lkeys = layer_keys(layers, dirs)
        self._sd = sd
        nbytes = lib().td_rnn_packed_bytes(C.byref(self.cfg))
        if nbytes == 0:
            raise _lib.TimediffError(lib().td_last_error().decode())
        self.packed = torch.empty(nbytes, dtype=torch.uint8, device=self.device)
        self._ws = None
        self._train_ws = None
        self._packT = None
        self._packT_fresh = False
        self.refresh()

    def refresh(self):
        """Re-read the parameters (call after every optimizer step): re-packs the MFMA weight cache."""
        self._lw = [_f32(self._sd[k]) for k in self._lkeys]
        self._misc = [_f32(self._sd[k]) for k in _lib.MISC_KEYS]
        self._misc_tab = ptr_table(self._misc)
        check(lib().td_rnn_pack_weights(C.byref(self.cfg), ptr_table(self._lw), ptr(self.packed), _stream()))
        self._packT_fresh = False

    @property
    def misc_table(self):
        return self._misc_tab

    def time_film(self, t, want_temb=False):
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        n = t.numel()
        film = torch.empty(n, 2 * self.cfg.dirs * self.cfg.hidden, dtype=torch.float32, device=self.device)
        temb = torch.empty(n, self.cfg.time_dim, dtype=torch.float32, device=self.device) if want_temb else None
        check(lib().td_rnn_time_film(C.byref(self.cfg), self._misc_tab, ptr(t), n, ptr(film), ptr(temb), _stream()))
        return (film, temb) if want_temb else film

    def time_film_backward(self, t, grad_film, sink=None):
        """Backward of time_film (the time-embedding MLP + FiLM linear, blocks.py:324-329 and :285): d film (n, 2DH) ->
        {misc key: grad} of the six parameters; sink: tensors to ACCUMULATE into instead (the trainer's flat buffer)."""
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        grad_film = _f32(grad_film)
        n = t.numel()
        keys = _lib.MISC_KEYS[:6]
        if sink is None:
            g = {k: torch.zeros_like(self._misc[i]) for i, k in enumerate(keys)}
        else:
            g = {k: sink[k] for k in keys}
            for i, k in enumerate(keys):
                assert g[k].dtype == torch.float32 and g[k].is_contiguous() and g[k].shape == self._misc[i].shape
        tab = (C.c_void_p * _lib.TD_MISC_COUNT)(*[g[k].data_ptr() if k in g else None for k in _lib.MISC_KEYS])
        ws = torch.empty(lib().td_rnn_time_film_backward_workspace_bytes(C.byref(self.cfg), n), dtype=torch.uint8, device=self.device)
        check(lib().td_rnn_time_film_backward(C.byref(self.cfg), self._misc_tab, ptr(t), n, ptr(grad_film), tab, ptr(ws), _stream()))
        return g

    def workspace(self, batch, seq_len):
        need = lib().td_rnn_workspace_bytes(C.byref(self.cfg), batch, seq_len)
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    def forward(self, x, t=None, film=None, want_hidden=False):
        """x: (B, c_in, L) fp32; t: (B,) int64 (or a precomputed film table (n, 2DH) with n in {1, B})."""
        x = _f32(x)
        B, _, L = x.shape
        if film is None:
            film = self.time_film(t)
        y = torch.empty(B, self.cfg.c_out, L, dtype=torch.float32, device=self.device)
        ws = self.workspace(B, L)
        check(lib().td_rnn_forward(C.byref(self.cfg), ptr(self.packed), self._misc_tab, ptr(x), ptr(film),
                                   film.shape[0], None, B, L, ptr(y), ptr(ws), _stream()))
        if want_hidden:
            hidden = torch.empty(B, L, self.cfg.dirs * self.cfg.hidden, dtype=torch.float32, device=self.device)
            check(lib().td_rnn_read_hidden(C.byref(self.cfg), ptr(ws), B, L, ptr(hidden), _stream()))
            return y, hidden
        return y


    # -- training -------------------------------------------------------------------------------
    @staticmethod
    def set_dropout(p, seed=0):
        """Inter-layer dropout of the NEXT forward_train / backward pair (process-wide, td_rnn_set_dropout)."""
        check(lib().td_rnn_set_dropout(float(p), int(seed)))

    def dropout_mask(self, batch, seq_len, layer):
        """(B, L, D*H) mask the current (p, seed) puts on layer `layer`'s output (parity checks)."""
        m = torch.empty(batch, seq_len, self.cfg.dirs * self.cfg.hidden, dtype=torch.float32, device=self.device)
        check(lib().td_rnn_dropout_mask(C.byref(self.cfg), batch, seq_len, layer, ptr(m), _stream()))
        return m

    def reserve_train(self, batch, seq_len):
        """Allocates what forward_train / backward of this shape keep between calls (the training workspace, the transposed
        weight cache): a caller that captures them in a graph does this first, outside the capture."""
        need = lib().td_rnn_train_workspace_bytes(C.byref(self.cfg), batch, seq_len)
        if need == 0:
            raise _lib.TimediffError(lib().td_last_error().decode() or "bad training shape")
        if self._train_ws is None or self._train_ws.numel() < need:
            self._train_ws = None
            self._train_ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        if self._packT is None:
            self._packT = torch.empty(lib().td_rnn_packT_bytes(C.byref(self.cfg)), dtype=torch.uint8, device=self.device)
        return self._train_ws, self._packT

    def forward_train(self, x, film):
        """Forward that keeps the activations for `backward` (one film row per batch element)."""
        x, film = _f32(x), _f32(film)
        B, _, L = x.shape
        assert film.shape == (B, 2 * self.cfg.dirs * self.cfg.hidden)
        self.reserve_train(B, L)
        y = torch.empty(B, self.cfg.c_out, L, dtype=torch.float32, device=self.device)
        check(lib().td_rnn_forward_train(C.byref(self.cfg), ptr(self.packed), self._misc_tab, ptr(x), ptr(film), B, L,
                                         ptr(y), ptr(self._train_ws), _stream()))
        self._train_shape = (B, L)
        return y

    def backward(self, grad_y, film, sink=None, on_layer_done=None, time=None):
        """Backward of the last forward_train.  -> (layer grads in layer_keys order, {misc key: grad}, grad_film).
        time: the (B,) int64 time steps the film rows were computed from: the time-embedding MLP's backward then runs inside
        the same call (td_rnn_backward_all: beside the BPTT sweeps) and {misc key: grad} covers all ten entries.
        sink: {parameter name: fp32 tensor} to ACCUMULATE the parameter gradients into (the trainer's flat gradient
        buffer) instead of fresh tensors; on_layer_done(l): called while the kernels are being enqueued, right after
        the last one that writes layer l's gradients (l = layers: the LayerNorm / fc gradients), see
        td_rnn_backward_cb."""
        B, L = self._train_shape
        grad_y, film = _f32(grad_y), _f32(film)
        if not self._packT_fresh:
            check(lib().td_rnn_pack_weights_T(C.byref(self.cfg), ptr_table(self._lw), ptr(self._packT), _stream()))
            self._packT_fresh = True
        first = 0 if time is not None else 6       # misc entries 0 .. 5: the time-embedding MLP and the FiLM linear
        if sink is None:
            g_layer = [torch.zeros_like(w) for w in self._lw]
            g_misc = {k: torch.zeros_like(self._misc[i]) for i, k in enumerate(_lib.MISC_KEYS) if i >= first}
        else:
            g_layer = [sink[k] for k in self._lkeys]
            g_misc = {k: sink[k] for i, k in enumerate(_lib.MISC_KEYS) if i >= first}
            for g, w in list(zip(g_layer, self._lw)) + [(g_misc[k], self._misc[i]) for i, k in enumerate(_lib.MISC_KEYS) if i >= first]:
                assert g.dtype == torch.float32 and g.is_contiguous() and g.shape == w.shape and g.device == w.device
        tab = (C.c_void_p * _lib.TD_MISC_COUNT)(*[g_misc[k].data_ptr() if k in g_misc else None for k in _lib.MISC_KEYS])
        g_film = torch.empty_like(film)
        failure = []

        def _done(layer, _user):
            try:
                on_layer_done(int(layer))
            except BaseException as e:      # an exception must not unwind through the C frames
                failure.append(e)

        cb = _lib.LAYER_DONE_FN(_done) if on_layer_done is not None else _lib.LAYER_DONE_FN()
        if time is not None:
            time = time.to(device=self.device, dtype=torch.int64).contiguous()
            assert time.numel() == B
            fws = torch.empty(lib().td_rnn_time_film_backward_workspace_bytes(C.byref(self.cfg), B), dtype=torch.uint8, device=self.device)
            check(lib().td_rnn_backward_all(C.byref(self.cfg), ptr(self._packT), self._misc_tab, ptr(film), ptr(time), ptr(grad_y), B, L,
                                            ptr(self._train_ws), ptr(fws), ptr_table(g_layer), tab, ptr(g_film), cb, None, _stream()))
        else:
            check(lib().td_rnn_backward_cb(C.byref(self.cfg), ptr(self._packT), self._misc_tab, ptr(film), ptr(grad_y), B, L,
                                           ptr(self._train_ws), ptr_table(g_layer), tab, ptr(g_film), cb, None, _stream()))
        if failure:
            raise failure[0]
        return g_layer, g_misc, g_film


class DiffusionKernels:
    """The MixedDiffusion arithmetic (mixed_diffusion.py) on the HIP kernels; `num_classes=[]` gives the
    Gaussian process of gaussian_diffusion.py."""

    def __init__(self, num_index, cat_index, num_classes, seq_len, timesteps, tables, data_channels, device, auto_normalize=True):
        self.n_num, self.n_cat = len(num_index), len(cat_index)
        self.K = [int(k) for k in num_classes]
        self.sumk = sum(self.K)
        self.L, self.T = seq_len, timesteps
        self.device = device
        self.data_channels = data_channels
        self.tables = torch.stack([tables[k].to(device=device, dtype=torch.float32) for k in _lib.TAB_ROWS]).contiguous()
        cfg = DiffusionConfig(n_num=self.n_num, n_cat=self.n_cat, seq_len=seq_len, timesteps=timesteps,
                              data_channels=data_channels, raw_range=0 if auto_normalize else 1, tables=self.tables.data_ptr())
        for i, k in enumerate(self.K):
            cfg.num_classes[i] = k
        for i, v in enumerate(num_index):
            cfg.num_index[i] = int(v)
        for i, v in enumerate(cat_index):
            cfg.cat_index[i] = int(v)
        self.cfg = cfg
        self._loss_ws = torch.empty(lib().td_mixed_loss_workspace_bytes(C.byref(cfg), 1), dtype=torch.uint8, device=device)

    def _rng(self, seed=0, patient0=0, tag=0, seed_dev=None):
        return Rng(seed=seed, patient0=patient0, tag=tag, seed_dev=None if seed_dev is None else seed_dev.data_ptr())

    def q_sample(self, data, t, noise=None, uniforms=None, seed=0, patient0=0, tag=0, seed_dev=None):
        """seed_dev: a device tensor whose first 8 bytes are added to `seed` when the kernel runs (captured training steps)."""
        data = _f32(data)
        B = data.shape[0]
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        x_in = torch.empty(B, self.n_num + self.sumk, self.L, dtype=torch.float32, device=self.device)
        x0_idx = torch.empty(B, self.n_cat, self.L, dtype=torch.uint8, device=self.device)
        xt_idx = torch.empty_like(x0_idx)
        noise_out = torch.empty(B, self.n_num, self.L, dtype=torch.float32, device=self.device)
        noise = None if noise is None else _f32(noise)
        uniforms = None if uniforms is None else _f32(uniforms)
        check(lib().td_q_sample_mixed(C.byref(self.cfg), B, ptr(data), ptr(t), ptr(noise), ptr(uniforms),
                                      self._rng(seed, patient0, tag, seed_dev), ptr(x_in), ptr(x0_idx), ptr(xt_idx), ptr(noise_out),
                                      _stream()))
        return x_in, x0_idx, xt_idx, noise_out

    def mixed_loss(self, model_out, noise, x0_idx, xt_idx, t, loss_lambda, scale=1.0, want_grad=True):
        model_out = _f32(model_out)
        B = model_out.shape[0]
        t = t.to(device=self.device, dtype=torch.int64).contiguous()
        loss = torch.empty(1, dtype=torch.float32, device=self.device)
        grad = torch.empty_like(model_out) if want_grad else None
        check(lib().td_mixed_loss(C.byref(self.cfg), B, ptr(model_out), ptr(noise), ptr(x0_idx), ptr(xt_idx), ptr(t),
                                  loss_lambda, scale, ptr(loss), ptr(grad), ptr(self._loss_ws), _stream()))
        return loss, grad

    def prior(self, B, noise=None, uniforms=None, seed=0, patient0=0):
        z = torch.empty(B, self.n_num, self.L, dtype=torch.float32, device=self.device)
        idx = torch.empty(B, self.n_cat, self.L, dtype=torch.uint8, device=self.device)
        noise = None if noise is None else _f32(noise)
        uniforms = None if uniforms is None else _f32(uniforms)
        check(lib().td_sample_prior(C.byref(self.cfg), B, ptr(noise), ptr(uniforms), self._rng(seed, patient0), ptr(z),
                                    ptr(idx), _stream()))
        return z, idx

    def build_input(self, z, idx):
        B = z.shape[0] if self.n_num else idx.shape[0]
        x_in = torch.empty(B, self.n_num + self.sumk, self.L, dtype=torch.float32, device=self.device)
        check(lib().td_build_denoiser_input(C.byref(self.cfg), B, ptr(z), ptr(idx), ptr(x_in), _stream()))
        return x_in

    def reverse_step_(self, model_out, z, idx, t, noise=None, uniforms=None, seed=0, patient0=0):
        """In place on z / idx."""
        model_out = _f32(model_out)
        B = model_out.shape[0]
        noise = None if noise is None else _f32(noise)
        uniforms = None if uniforms is None else _f32(uniforms)
        check(lib().td_reverse_step_mixed(C.byref(self.cfg), B, ptr(model_out), ptr(z), ptr(idx), int(t), None,
                                          ptr(noise), ptr(uniforms), self._rng(seed, patient0), _stream()))

    def reverse_step_terms(self, model_out, z, idx, t):
        """-> (x0, mean, log_theta) of the step as the fused kernel computes them (parity aid, nothing is sampled)."""
        model_out = _f32(model_out)
        B = model_out.shape[0]
        x0 = torch.empty(B, self.n_num, self.L, dtype=torch.float32, device=self.device)
        mean = torch.empty_like(x0)
        lth = torch.empty(B, self.sumk, self.L, dtype=torch.float32, device=self.device)
        check(lib().td_reverse_step_terms(C.byref(self.cfg), B, ptr(model_out), ptr(z), ptr(idx), int(t), ptr(x0), ptr(mean),
                                          ptr(lth), _stream()))
        return x0, mean, lth

    def decode(self, z, idx):
        B = z.shape[0] if self.n_num else idx.shape[0]
        out = torch.empty(B, self.data_channels, self.L, dtype=torch.float32, device=self.device)
        check(lib().td_decode_samples(C.byref(self.cfg), B, ptr(z), ptr(idx), ptr(out), _stream()))
        return out


class SamplerKernels:
    """The full reverse chain (MixedDiffusion.sample) as replayed hipGraph steps."""

    def __init__(self, denoiser: DenoiserKernels, diffusion: DiffusionKernels):
        self.den, self.diff = denoiser, diffusion
        self._stream = torch.cuda.Stream(device=denoiser.device)
        self._film = None
        self._ws = None

    def refresh(self):
        """Recompute the (T, 2DH) FiLM table (call after the parameters changed)."""
        t = torch.arange(self.diff.T, device=self.den.device, dtype=torch.int64)
        self._film = self.den.time_film(t)

    def sample(self, batch, noise=None, uniforms=None, seed=0, patient0=0, use_graph=True):
        """-> (B, data_channels, L) fp32 on the device.  noise: (T+1, B, n_num, L), uniforms: (T+1, B, sum K, L)
        in loop order (slab 0 = prior draw, slab 1+s = step t = T-1-s), or None for the in-kernel Philox."""
        if self._film is None:
            self.refresh()
        need = lib().td_sample_chain_workspace_bytes(C.byref(self.den.cfg), C.byref(self.diff.cfg), batch)
        if need == 0:
            raise _lib.TimediffError(lib().td_last_error().decode() or "bad sampler configuration")
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.den.device)
        out = torch.empty(batch, self.diff.data_channels, self.diff.L, dtype=torch.float32, device=self.den.device)
        noise = None if noise is None else _f32(noise)
        uniforms = None if uniforms is None else _f32(uniforms)
        self._stream.wait_stream(torch.cuda.current_stream())
        check(lib().td_sample_chain(C.byref(self.den.cfg), C.byref(self.diff.cfg), ptr(self.den.packed),
                                    self.den.misc_table, ptr(self._film), batch, ptr(noise), ptr(uniforms),
                                    Rng(seed=seed, patient0=patient0, tag=0), int(use_graph), ptr(out), ptr(self._ws),
                                    C.c_void_p(self._stream.cuda_stream)))
        torch.cuda.current_stream().wait_stream(self._stream)
        return out

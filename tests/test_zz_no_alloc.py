"""The one entry point that allocates nothing: td_mixed_loss_noalloc (include/timediff_hip.h) against
td_mixed_loss, through the C ABI, on two calls in a row, the second with a smaller input (the file-scope scratch
keeps the first call's partials).  Collected last (file name): a failure here must not hide other tests."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _case(dk, B, L, n_num, K, T, seed, dev):
    g = torch.Generator().manual_seed(seed)
    sumk = sum(K)
    mo = torch.randn(B, n_num + sumk, L, generator=g).to(dev)
    noise = torch.randn(B, n_num, L, generator=g).to(dev)
    x0 = torch.stack([torch.randint(0, k, (B, L), generator=g) for k in K], 1).to(torch.uint8).to(dev)
    xt = torch.stack([torch.randint(0, k, (B, L), generator=g) for k in K], 1).to(torch.uint8).to(dev)
    t = torch.randint(0, T, (B,), generator=g).to(dev)
    t[0] = 0
    return mo, noise, x0, xt, t


def test_mixed_loss_noalloc_matches_workspace_variant():
    from oracle import oracle as orc
    from timediff_amd import _lib
    from timediff_amd._lib import check, lib, ptr
    from timediff_amd.ops import DiffusionKernels
    assert torch.cuda.is_available(), "this test needs the MI355X"
    dev = torch.device("cuda:0")
    L, T, K, num, cat = 272, 1000, [2, 2, 3, 2, 2], [0, 2, 4, 6], [1, 3, 5, 7, 8]
    tables = {k: torch.from_numpy(v) for k, v in orc.schedule(T, "cosine").items()}
    dk = DiffusionKernels(num, cat, K, L, T, tables, 9, dev)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for B, seed in ((1500, 1), (37, 2)):       # 1500 x 272 positions fill all 1024 partial slots; then 40 blocks
        mo, noise, x0, xt, t = _case(dk, B, L, len(num), K, T, seed, dev)
        want_loss, want_grad = dk.mixed_loss(mo, noise, x0, xt, t, 0.8, scale=0.5)
        loss = torch.full((1,), float("nan"), device=dev)
        grad = torch.full_like(mo, float("nan"))
        check(lib().td_mixed_loss_noalloc(C.byref(dk.cfg), B, ptr(mo), ptr(noise), ptr(x0), ptr(xt), ptr(t), 0.8, 0.5,
                                          ptr(loss), ptr(grad), stream))
        torch.cuda.synchronize()
        # same kernels, same partial order: bit-exact
        assert torch.equal(loss, want_loss), (float(loss), float(want_loss))
        assert torch.equal(grad, want_grad)
        ref = orc.Diffusion(len(num), K, L, T).mixed_loss(mo.cpu().numpy(), noise.cpu().numpy(), x0.cpu().numpy(),
                                                           xt.cpu().numpy(), t.cpu().numpy(), 0.8)
        np.testing.assert_allclose(float(loss), ref["loss"], rtol=2e-5)
    # beyond the stated maximum: a clear error, nothing launched
    rc = lib().td_mixed_loss_noalloc(C.byref(dk.cfg), (1 << 27) // L + 1, ptr(mo), ptr(noise), ptr(x0), ptr(xt), ptr(t),
                                     0.8, 1.0, ptr(loss), None, stream)
    assert rc == 3 and b"exceeds the maximum" in lib().td_last_error()

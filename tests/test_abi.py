"""The C-ABI library loads and exports every symbol include/timediff_hip.h declares (no compute)."""
import ctypes
import os
import re

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "timediff_hip.h")).read()
    return sorted(set(re.findall(r"TD_API[^;(]*?\b(td_\w+)\s*\(", text)))


def test_header_declares_entry_points():
    names = declared_symbols()
    for must in ("td_rnn_forward", "td_reverse_step_mixed", "td_mixed_loss", "td_sample_chain", "td_q_sample_mixed"):
        assert must in names


def test_library_exports_every_declared_symbol():
    from timediff_amd import _lib
    handle = ctypes.CDLL(_lib.LIB_PATH)
    for name in declared_symbols():
        assert hasattr(handle, name), f"{name} declared in timediff_hip.h but not exported"
    assert set(_lib._SIGS) == set(declared_symbols())
    assert _lib.lib().td_version() >= 100


def test_argument_validation_without_gpu():
    """Entry points reject bad configurations before touching the device."""
    from timediff_amd import _lib
    L = _lib.lib()
    cfg = _lib.RnnConfig(c_in=14, hidden=4096, c_out=14, layers=3, dirs=2, gru=0, embed_dim=64, time_dim=256)
    assert L.td_rnn_packed_bytes(ctypes.byref(cfg)) == 0
    assert b"hidden" in L.td_last_error()
    cfg.hidden = 256
    assert L.td_rnn_packed_bytes(ctypes.byref(cfg)) > 0
    assert L.td_rnn_workspace_bytes(ctypes.byref(cfg), 100, 272) > 0
    cfg.dirs = 3
    assert L.td_rnn_packed_bytes(ctypes.byref(cfg)) == 0


def test_kernel_selection_knobs_without_gpu():
    """The launch-form queries are host arithmetic: which recurrent-layer family AUTO takes and whether it launches step by
    step (td_set_step_launch), per width and batch; unknown knob values are refused."""
    from timediff_amd import _lib
    L = _lib.lib()

    def choice(hidden, batch, seq_len=272):
        cfg = _lib.RnnConfig(c_in=14, hidden=hidden, c_out=14, layers=3, dirs=2, gru=0, embed_dim=64, time_dim=256)
        return L.td_rnn_layer_kernel_choice(ctypes.byref(cfg), batch, seq_len), L.td_rnn_step_launch_choice(ctypes.byref(cfg), batch, seq_len)

    assert L.td_set_step_launch(7) != 0 and b"td_set_step_launch" in L.td_last_error()
    assert L.td_set_layer_kernel(9) != 0
    try:
        assert choice(256, 64) == (_lib.LAYER_KERNEL_CLUSTER, 0)           # the reference's training batch
        assert choice(256, 16384) == (_lib.LAYER_KERNEL_BATCH, 0)          # bench.py's sampler batch
        assert choice(92, 256) == (_lib.LAYER_KERNEL_CLUSTER, 0)           # MIMIC-III width: the one-workgroup form
        assert choice(512, 512) == (_lib.LAYER_KERNEL_UNITS, 1)            # configs[4] per GPU: one launch per time step
        assert choice(1024, 2048) == (_lib.LAYER_KERNEL_UNITS, 1)          # 128 (batch block, direction) pairs: still per step
        assert choice(512, 8192) == (_lib.LAYER_KERNEL_UNITS, 0)           # configs[3]: the persistent kernel
        assert L.td_set_step_launch(_lib.STEP_LAUNCH_OFF) == 0
        assert choice(512, 512) == (_lib.LAYER_KERNEL_UNITS, 0)
        assert L.td_set_step_launch(_lib.STEP_LAUNCH_ON) == 0
        assert choice(512, 8192) == (_lib.LAYER_KERNEL_UNITS, 1)
        assert choice(192, 64) == (_lib.LAYER_KERNEL_UNITS, 0)             # 6 unit blocks: not whole groups of four
        assert choice(128, 64) == (_lib.LAYER_KERNEL_UNITS, 1)
        assert choice(320, 64) == (_lib.LAYER_KERNEL_UNITS, 1)             # 10 unit blocks run as 12
    finally:
        assert L.td_set_step_launch(_lib.STEP_LAUNCH_AUTO) == 0
    assert choice(0, 64) == (-1, -1)


def _diff_cfg(_lib, **over):
    """An eICU-shaped td_diffusion_config whose `tables` pointer is a dummy: every call below must be
    rejected before anything dereferences it."""
    cfg = _lib.DiffusionConfig()
    cfg.n_num, cfg.n_cat, cfg.seq_len, cfg.timesteps = 4, 5, 272, 1000
    for v in range(5):
        cfg.num_classes[v] = 2
        cfg.cat_index[v] = 4 + v
    for i in range(4):
        cfg.num_index[i] = i
    cfg.data_channels, cfg.raw_range, cfg.tables = 9, 0, 0x1000
    for k, v in over.items():
        if isinstance(v, tuple):
            getattr(cfg, k)[v[0]] = v[1]
        else:
            setattr(cfg, k, v)
    return cfg


def test_diffusion_entry_points_reject_bad_arguments_without_gpu():
    """The error behaviour of the diffusion entry points (include/timediff_hip.h: TD_ERR_INVALID for bad
    arguments, TD_ERR_UNSUPPORTED beyond the built limits), checked through the C-ABI: each call returns before
    any device work, so this runs without a GPU."""
    from timediff_amd import _lib
    L = _lib.lib()
    P = 0x1000      # non-NULL dummy pointer; never dereferenced by a rejected call
    rng = _lib.Rng(1, 0, 0)
    INVALID, UNSUPPORTED = 1, 3

    def q_sample(cfg, batch=8, data=P, t=P, x_in=P, x0=P, xt=P):
        return L.td_q_sample_mixed(ctypes.byref(cfg) if cfg is not None else None, batch, data, t, None, None, rng, x_in, x0, xt,
                                   None, None)

    assert q_sample(None) == INVALID and b"NULL" in L.td_last_error()
    assert q_sample(_diff_cfg(_lib), batch=0) == INVALID and b"batch" in L.td_last_error()
    assert q_sample(_diff_cfg(_lib), batch=-3) == INVALID
    assert q_sample(_diff_cfg(_lib, seq_len=0)) == INVALID
    assert q_sample(_diff_cfg(_lib, timesteps=0)) == INVALID
    assert q_sample(_diff_cfg(_lib, tables=None)) == INVALID and b"tables" in L.td_last_error()
    assert q_sample(_diff_cfg(_lib, n_num=0, n_cat=0)) == INVALID and b"no channels" in L.td_last_error()
    assert q_sample(_diff_cfg(_lib, n_cat=_lib.TD_MAX_CAT + 1)) == INVALID and b"n_cat" in L.td_last_error()
    assert q_sample(_diff_cfg(_lib, num_classes=(2, 0))) == UNSUPPORTED and b"num_classes[2]" in L.td_last_error()
    assert q_sample(_diff_cfg(_lib, num_classes=(1, 10 ** 6))) == UNSUPPORTED
    assert q_sample(_diff_cfg(_lib, num_index=(3, 9))) == INVALID and b"num_index[3]" in L.td_last_error()
    assert q_sample(_diff_cfg(_lib, cat_index=(0, -1))) == INVALID and b"cat_index[0]" in L.td_last_error()
    assert q_sample(_diff_cfg(_lib), data=None) == INVALID
    assert q_sample(_diff_cfg(_lib), x0=None) == INVALID and b"x0_idx" in L.td_last_error()

    def reverse(cfg, t_host, t_dev=None, mo=P, z=P, idx=P):
        return L.td_reverse_step_mixed(ctypes.byref(cfg), 8, mo, z, idx, t_host, t_dev, None, None, rng, None)

    assert reverse(_diff_cfg(_lib), 1000) == INVALID and b"outside [0, 1000)" in L.td_last_error()
    assert reverse(_diff_cfg(_lib), -1) == INVALID
    assert reverse(_diff_cfg(_lib), 5, mo=None) == INVALID
    assert reverse(_diff_cfg(_lib), 5, idx=None) == INVALID
    assert reverse(_diff_cfg(_lib, n_cat=0), 1000, idx=None) == INVALID and b"outside" in L.td_last_error()

    assert L.td_mixed_loss_workspace_bytes(ctypes.byref(_diff_cfg(_lib)), 64) > 0


def test_rnn_entry_points_reject_bad_arguments_without_gpu():
    from timediff_amd import _lib
    L = _lib.lib()
    P = 0x1000
    cfg = _lib.RnnConfig(c_in=14, hidden=256, c_out=14, layers=3, dirs=2, gru=0, embed_dim=64, time_dim=256)
    misc = (ctypes.c_void_p * 16)()

    def forward(batch=8, seq_len=272, film_rows=1, film_row=None, x=P, y=P, ws=P):
        return L.td_rnn_forward(ctypes.byref(cfg), P, misc, x, P, film_rows, film_row, batch, seq_len, y, ws, None)

    assert forward(batch=0) == 1 and b"positive" in L.td_last_error()
    assert forward(seq_len=0) == 1
    assert forward(x=None) == 1 and b"NULL" in L.td_last_error()
    assert forward(ws=None) == 1
    assert forward(film_rows=3) == 1 and b"film_rows" in L.td_last_error()
    for field, bad in (("layers", 0), ("layers", 99), ("c_in", 0), ("c_in", 4096), ("c_out", 0), ("c_out", 4096), ("hidden", 0)):
        good = getattr(cfg, field)
        setattr(cfg, field, bad)
        assert L.td_rnn_packed_bytes(ctypes.byref(cfg)) == 0, (field, bad)
        assert L.td_rnn_workspace_bytes(ctypes.byref(cfg), 64, 272) == 0, (field, bad)
        assert L.td_rnn_train_workspace_bytes(ctypes.byref(cfg), 64, 272) == 0, (field, bad)
        setattr(cfg, field, good)
    assert L.td_rnn_workspace_bytes(ctypes.byref(cfg), 0, 272) == 0
    assert L.td_rnn_workspace_bytes(ctypes.byref(cfg), 64, 0) == 0
    assert L.td_rnn_set_dropout(1.0, 0) == 1 and L.td_rnn_set_dropout(-0.1, 0) == 1
    assert L.td_rnn_set_dropout(0.0, 0) == 0

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

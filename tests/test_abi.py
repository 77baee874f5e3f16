"""CPU tests: the C-ABI library loads without a GPU and exports every symbol include/kd_hip.h
declares (no compute calls), argument validation returns negative codes with a message, and the
host-side queries agree with the kernels' launch geometry."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported():
    from kdrt.lib import HEADER_PATH, SO_PATH, parse_header
    protos = parse_header(HEADER_PATH)
    text = open(HEADER_PATH).read()
    declared = set(re.findall(r"\b(kd_[a-z0-9_]+)\s*\(", re.sub(r"/\*.*?\*/", "", text, flags=re.S)))
    assert declared == set(protos), declared ^ set(protos)
    assert len(protos) >= 40
    dll = ctypes.CDLL(SO_PATH)
    for name in protos:
        assert hasattr(dll, name), f"{name} declared in kd_hip.h but not exported by libkd_hip.so"


def test_version_arch_and_queries():
    from kdrt.lib import lib
    assert lib.kd_version() >= 100
    assert lib.kd_arch() == b"gfx950"
    assert lib.kd_pwconv_stat_rows(128) == 1 and lib.kd_pwconv_stat_rows(129) == 2
    assert lib.kd_pwconv_wgrad_ws_bytes(1 << 20, 768, 128) >= 768 * 128 * 4
    assert lib.kd_rowwise_stat_rows(10, 128) == 2          # 8 row slots per 256-thread block at C=128
    assert lib.kd_rowwise_stat_rows(1 << 30, 128) == 2048   # capped grid
    assert lib.kd_seg_loss_ws_bytes(4096) > 0 and lib.kd_mse_ws_bytes(1 << 20) > 0


def test_argument_errors_are_reported_not_thrown():
    from kdrt.lib import KDError, lib
    rc = lib.kd_pwconv_gemm(None, 0, None, 0, 0, 0, None, None, None, None, None, None, None, None, 0, None, 0, 0, None, 0,
                            None, None, None, None, 0, None, 0, 0, 0, 0, None, None)
    assert rc < 0
    assert b"kd_pwconv_gemm" in lib.kd_last_error_string()
    with pytest.raises(KDError):
        lib.call("kd_transpose", None, None, 0, 0, None)


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    import sys
    import kdrt  # noqa: F401
    L = sys.modules["kdrt.lib"]          # (the package attribute `kdrt.lib` is the bound library object)
    monkeypatch.setattr(L, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.KDError):
        L._Lib()

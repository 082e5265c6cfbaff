"""The C-ABI library loads and exports every symbol include/pings_hip.h declares."""
import ctypes

from pings_amd import _lib


def test_header_declares_symbols():
    names = _lib.header_symbols()
    assert "pings_abi_version" in names and "pings_ssim_forward" in names
    assert len(names) == len(set(names))


def test_library_exports_every_declared_symbol():
    L = _lib.lib()
    missing = [n for n in _lib.header_symbols() if not hasattr(L, n)]
    assert not missing, f"symbols declared in include/pings_hip.h but not exported: {missing}"
    assert L.pings_abi_version() == _lib.expected_abi() >= 7
    assert L.pings_last_error() is not None


def test_argument_errors_are_status_codes_not_crashes():
    L = _lib.lib()
    # null pointers / empty shapes are rejected before any GPU work is attempted
    st = L.pings_ssim_forward(None, None, 3, 8, 8, 0, None, None, None, None, None, None)
    assert st == 1
    assert b"null" in L.pings_last_error()
    assert L.pings_ssim_partials_count(0, 8, 8) == 0


def test_a_library_of_another_abi_version_is_refused(monkeypatch):
    """a stale or A/B library must not be called with this package's argtypes."""
    import pytest

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "expected_abi", lambda: 10_000)
    with pytest.raises(_lib.PingsHipError, match="ABI version"):
        _lib.lib()
    monkeypatch.undo()
    assert _lib.lib().pings_abi_version() == _lib.expected_abi()


def test_double_backward_entry_rejects_other_shapes_before_any_gpu_work():
    """`pings_mlp_double_backward` covers the SDF decoder shape (hidden 64, one output); anything else is a status code
    (the host wrapper then composes the node from device operators), decided before a single HIP call."""
    L = _lib.lib()
    L.pings_mlp_double_backward_supported.restype = ctypes.c_int
    L.pings_mlp_double_backward_supported.argtypes = [ctypes.c_int] * 3
    assert L.pings_mlp_double_backward_supported(35, 64, 1) == 1
    assert L.pings_mlp_double_backward_supported(11, 64, 1) == 1
    assert L.pings_mlp_double_backward_supported(32, 128, 24) == 0
    assert L.pings_mlp_double_backward_supported(35, 64, 3) == 0
    vp, i64, i32 = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int
    L.pings_mlp_double_backward.restype = ctypes.c_int
    L.pings_mlp_double_backward.argtypes = [vp, vp, vp, i64, i32, i32, i32, vp, vp, vp, vp, vp, vp, vp, vp]
    st = L.pings_mlp_double_backward(None, None, None, 10, 32, 128, 24, None, None, None, None, None, None, None, None)
    assert st == 1 and b"double backward" in L.pings_last_error()
    st = L.pings_mlp_double_backward(None, None, None, 10, 35, 64, 1, None, None, None, None, None, None, None, None)
    assert st == 1 and b"null" in L.pings_last_error()

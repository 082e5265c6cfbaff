"""ctypes binding of libpings_hip.so (the C ABI declared in include/pings_hip.h).

The product path has no CPU fallback: if the library is missing, or a call is
made with tensors that are not on a HIP device, this module raises.
"""
from __future__ import annotations

import ctypes as C
import re
from pathlib import Path

import torch

PKG = Path(__file__).resolve().parent
# PINGS_HIP_LIB: another build of the same library (A/B timing of kernel variants; never set in production)
import os as _os

LIB_PATH = Path(_os.environ["PINGS_HIP_LIB"]) if _os.environ.get("PINGS_HIP_LIB") else PKG / "lib" / "libpings_hip.so"
if _os.environ.get("PINGS_HIP_LIB"):
    import warnings as _warnings

    _warnings.warn(f"pings_amd: loading the HIP library from PINGS_HIP_LIB={LIB_PATH} (A/B timing only)")
HEADER = PKG.parent / "include" / "pings_hip.h"

_lib = None

c_fp = C.c_void_p  # device pointers travel as integers (tensor.data_ptr())


class PingsHipError(RuntimeError):
    pass


def header_symbols() -> list[str]:
    """Names of every PINGS_API function declared in include/pings_hip.h."""
    txt = HEADER.read_text()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return re.findall(r"PINGS_API\s+[\w\s\*]+?\b(pings_\w+)\s*\(", txt)


def expected_abi() -> int:
    """PINGS_ABI_VERSION of the header this package's ctypes signatures were written against."""
    m = re.search(r"#define\s+PINGS_ABI_VERSION\s+(\d+)", HEADER.read_text())
    if not m:
        raise PingsHipError(f"{HEADER} does not define PINGS_ABI_VERSION")
    return int(m.group(1))


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise PingsHipError(
                f"{LIB_PATH} is missing: build it with `python -m pings_amd.build` "
                "(or __graft_entry__.build()). There is no CPU fallback for the HIP path.")
        l = C.CDLL(str(LIB_PATH))
        _declare(l)
        # a stale libpings_hip.so (or a PINGS_HIP_LIB A/B build of another revision) would be called with this
        # package's argtypes: refuse it here instead of passing mismatched arguments into kernels
        got, want = l.pings_abi_version(), expected_abi()
        if got != want:
            raise PingsHipError(f"{LIB_PATH} reports ABI version {got}, this package binds version {want} "
                                f"(include/pings_hip.h): rebuild with `python -m pings_amd.build`")
        _lib = l
    return _lib


def _declare(l: C.CDLL) -> None:
    l.pings_abi_version.restype = C.c_int
    l.pings_last_error.restype = C.c_char_p
    l.pings_ssim_partials_count.restype = C.c_size_t
    l.pings_ssim_partials_count.argtypes = [C.c_int] * 3
    l.pings_ssim_forward.restype = C.c_int
    l.pings_ssim_forward.argtypes = [c_fp, c_fp, C.c_int, C.c_int, C.c_int, C.c_int,
                                     c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]
    l.pings_ssim_backward.restype = C.c_int
    l.pings_ssim_backward.argtypes = [c_fp, c_fp, C.c_int, C.c_int, C.c_int,
                                      c_fp, c_fp, c_fp, c_fp, c_fp, c_fp]


def check(status: int, what: str) -> None:
    if status != 0:
        msg = lib().pings_last_error().decode(errors="replace")
        raise PingsHipError(f"{what} failed with status {status}: {msg}")


def ptr(t: torch.Tensor | None):
    """Device pointer of a contiguous HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise PingsHipError(
            "pings_amd ops run on the HIP device only (got a CPU tensor); "
            "there is no CPU fallback")
    if not t.is_contiguous():
        raise PingsHipError("pings_amd internal error: non-contiguous tensor at the C ABI")
    return t.data_ptr()


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def stream_ptr(device=None):
    """`hipStream_t` of torch's current stream on `device` as an integer (a dozen calls per training step: the raw
    getter costs 0.3 us against 4 us for building a `torch.cuda.Stream` object each time)."""
    if _raw_stream is not None:
        idx = getattr(device, "index", None)
        if idx is None:
            idx = device if isinstance(device, int) else torch.cuda.current_device()
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


# ---------------------------------------------------------------- host round trips
# Every place where the host waits for a device value (a D2H read-back) calls note_sync(), so that bench.py can report
# `host_syncs_per_render` and tests can pin the count.  The reference's `render` has three of its own
# (gaussian_renderer/__init__.py:219 `.item()`, :305 NaN assert, and the boolean-mask compaction in spawn :730-737).
SYNC_LOG: dict[str, int] = {}


def note_sync(what: str) -> None:
    SYNC_LOG[what] = SYNC_LOG.get(what, 0) + 1


def sync_counts(reset: bool = False) -> dict[str, int]:
    out = dict(SYNC_LOG)
    if reset:
        SYNC_LOG.clear()
    return out


def host_values(t: torch.Tensor) -> list:
    """Values of a small settings tensor on the host.  A tensor built from host numbers carries them as
    `_pings_host` (`with_host_values`); otherwise one D2H copy is made and remembered on the tensor object, so a
    persistent camera tensor pays it once, not once per frame."""
    held = getattr(t, "_pings_host", None)
    if held is not None and held[0] == t._version:
        return held[1]
    note_sync("settings_tensor_readback")       # first use, or the tensor was updated in place since the last read
    hv = t.detach().to("cpu", torch.float32).tolist()
    try:
        t._pings_host = (t._version, hv)
    except Exception:
        pass
    return hv


def with_host_values(t: torch.Tensor, values) -> torch.Tensor:
    t._pings_host = (t._version, [float(v) for v in values])
    return t

"""ctypes binding of libs2k.so (include/s2k.h).  There is no fallback: if the HIP library is not
built, or a tensor is not on the GPU, the product path raises."""
from __future__ import annotations

import ctypes
import os
from pathlib import Path

import numpy as np
import torch  # noqa: F401  -- BEFORE libs2k.so is loaded: PyTorch-ROCm bundles its own libamdhip64.so.7 / libhsa-runtime64; dlopen-ing
#                        libs2k.so first would bind the process to /opt/rocm's copies (same SONAME) and leave two HIP runtimes that
#                        do not see each other's device state (launches then fail with "no ROCm-capable device is detected")

from .plan import opdefs as D
from .plan.program import OP_DTYPE

_HERE = Path(__file__).resolve().parent
LIB_PATH = _HERE / "libs2k.so"
_lib = None


class S2kError(RuntimeError):
    pass


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    path = Path(os.environ.get("S2K_LIB", LIB_PATH))
    if not path.exists():
        raise S2kError(f"{path} not found: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'); "
                       "there is no CPU fallback for this path")
    L = ctypes.CDLL(str(path))
    L.s2k_abi_version.restype = ctypes.c_int
    L.s2k_op_size.restype = ctypes.c_size_t
    L.s2k_last_error.restype = ctypes.c_char_p
    L.s2k_kind_name.restype = ctypes.c_char_p
    L.s2k_kind_name.argtypes = [ctypes.c_int]
    vp, i32 = ctypes.c_void_p, ctypes.c_int
    L.s2k_program_run.restype = i32
    L.s2k_program_run.argtypes = [vp, i32, i32, vp, i32, vp]
    L.s2k_op_launch.restype = i32
    L.s2k_op_launch.argtypes = [vp, vp, i32, vp]
    L.s2k_program_profile.restype = i32
    L.s2k_program_profile.argtypes = [vp, i32, i32, vp, i32, vp, vp, vp]
    L.s2k_program_profile_ops.restype = i32
    L.s2k_program_profile_ops.argtypes = [vp, i32, i32, vp, i32, vp, vp]
    L.s2k_program_profile_variants.restype = i32
    L.s2k_program_profile_variants.argtypes = [vp, i32, i32, vp, i32, vp, vp, vp]
    f64 = ctypes.c_double
    L.s2k_adam_step.restype = i32
    L.s2k_adam_step.argtypes = [vp, vp, vp, vp, ctypes.c_int64, f64, f64, f64, f64, f64, i32, vp]
    L.s2k_measure_peaks.restype = i32
    L.s2k_measure_peaks.argtypes = [vp, ctypes.c_size_t, i32, vp, vp, vp, vp]
    L.s2k_selftest_mfma.restype = i32
    L.s2k_selftest_mfma.argtypes = [vp, vp, vp, vp]
    if L.s2k_abi_version() != 2:
        raise S2kError(f"libs2k ABI {L.s2k_abi_version()} != 2")
    if L.s2k_op_size() != D.OP_BYTES or OP_DTYPE.itemsize != D.OP_BYTES:
        raise S2kError("S2kOp layout mismatch between Python and libs2k")
    for name, kind in D.KIND.items():
        got = L.s2k_kind_name(kind)
        if got is None or got.decode() != name:
            raise S2kError(f"stage kind table mismatch at {name}")
    _lib = L
    return L


def check(rc: int) -> None:
    if rc != 0:
        raise S2kError(f"s2k error {rc}: {lib().s2k_last_error().decode()}")


class Bases:
    """The `void* bases[]` array handed to s2k_program_run."""

    def __init__(self):
        self.arr = (ctypes.c_void_p * len(D.BASES))()
        self.keep = {}
        self.device = None      # device of the tensors behind the bases: made current around every launch (run / profile)

    def set(self, name: str, tensor) -> "Bases":
        self.keep[name] = tensor
        if tensor is not None and self.device is None and tensor.is_cuda:
            self.device = tensor.device
        self.arr[D.BASE[name]] = None if tensor is None else tensor.data_ptr()
        return self

    @property
    def ptr(self):
        return ctypes.cast(self.arr, ctypes.c_void_p)


def _guard(bases: Bases):
    """The default stream's handle is 0 on every device, so the library cannot tell the device from the stream alone:
    make the device that owns the tensors current for the call (a module on cuda:1 with cuda:0 current otherwise launches
    on the wrong device)."""
    import contextlib

    return torch.cuda.device(bases.device) if bases.device is not None else contextlib.nullcontext()


def run(packed: np.ndarray, bases: Bases, stream: int, begin: int = 0, end: int | None = None) -> None:
    end = len(packed) if end is None else end
    with _guard(bases):
        check(lib().s2k_program_run(packed.ctypes.data, begin, end, bases.ptr, len(D.BASES), stream))


def profile(packed: np.ndarray, bases: Bases, stream: int):
    nk = len(D.OPS) + 1
    ms = np.zeros(nk, dtype=np.float32)
    cnt = np.zeros(nk, dtype=np.int32)
    with _guard(bases):
        check(lib().s2k_program_profile(packed.ctypes.data, 0, len(packed), bases.ptr, len(D.BASES), stream,
                                        ms.ctypes.data, cnt.ctypes.data))
    names = {v: k for k, v in D.KIND.items()}
    return {names[k]: (float(ms[k]), int(cnt[k])) for k in range(1, nk) if cnt[k]}


def profile_ops(packed: np.ndarray, bases: Bases, stream: int) -> np.ndarray:
    """Per-stage device milliseconds (HIP events on the launch stream)."""
    ms = np.zeros(len(packed), dtype=np.float32)
    with _guard(bases):
        check(lib().s2k_program_profile_ops(packed.ctypes.data, 0, len(packed), bases.ptr, len(D.BASES), stream, ms.ctypes.data))
    return ms


def profile_variants(packed: np.ndarray, bases: Bases, stream: int):
    """(ms per stage, variant per stage): variant 1 = the producer/consumer kernel was launched (include/s2k.h)."""
    ms = np.zeros(len(packed), dtype=np.float32)
    var = np.zeros(len(packed), dtype=np.int32)
    with _guard(bases):
        check(lib().s2k_program_profile_variants(packed.ctypes.data, 0, len(packed), bases.ptr, len(D.BASES), stream,
                                                 ms.ctypes.data, var.ctypes.data))
    return ms, var


def measure_peaks(device=None, waves_per_simd: int = 1, scratch_gib: float = 2.0) -> dict:
    """{"mfma_f32_tflops", "mfma_clock_mhz", "copy_gbps"} measured on `device` (include/s2k.h, s2k_measure_peaks)."""
    dev = torch.device(device if device is not None else "cuda")
    buf = torch.empty(int(scratch_gib * (1 << 30)), dtype=torch.uint8, device=dev)
    buf.view(torch.float32)[: (1 << 22)].normal_()
    out = (ctypes.c_double * 3)()
    addr = ctypes.addressof(out)
    with torch.cuda.device(dev):
        check(lib().s2k_measure_peaks(buf.data_ptr(), buf.numel(), waves_per_simd, addr, addr + 8, addr + 16,
                                      torch.cuda.current_stream(dev).cuda_stream))
    return {"mfma_f32_tflops": out[0], "mfma_clock_mhz": out[1], "copy_gbps": out[2]}

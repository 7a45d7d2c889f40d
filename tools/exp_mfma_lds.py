"""Tuning build only: rate of the f32 MFMA with 0 / 3 / 5 / 9 LDS operand reads per 9 MFMAs, 1 or 2 waves per SIMD."""
import ctypes
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

L = ctypes.CDLL(str(ROOT / "sentinel2-landcover-classification_amd" / "libs2k_tuning.so"))
L.s2k_measure_mfma_lds.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
buf = torch.zeros(1 << 20, device="cuda")
out = ctypes.c_double()
for w in (1, 2):
    for r in (0, 3, 5, 9):
        L.s2k_measure_mfma_lds(buf.data_ptr(), r, w, ctypes.addressof(out), torch.cuda.current_stream().cuda_stream)
        print(f"waves/SIMD {w}  LDS reads per 9 MFMAs {r}: {out.value:.1f} TF/s")

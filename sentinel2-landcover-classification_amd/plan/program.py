"""Program builder: tensor references, a bump arena, and packing of stage records.

The packed array is what `s2k_program_run` (include/s2k.h) consumes.
"""
from __future__ import annotations

from dataclasses import dataclass, replace

import numpy as np

from . import opdefs as D

ITEMSIZE = {"f32": 4, "f64": 8, "i64": 8, "i32": 4, "i16": 2, "u8": 1, "bf16": 2}
ALIGN = 256

OP_DTYPE = np.dtype([("kind", "<i4"), ("flags", "<i4"), ("t", "<i8", (D.N_T,)), ("n", "<i8", (D.N_N,)),
                     ("d", "<i4", (D.N_D,)), ("f", "<f4", (D.N_F,))])
assert OP_DTYPE.itemsize == D.OP_BYTES


@dataclass(frozen=True)
class TRef:
    """A tensor inside one of the base allocations: byte offset + logical shape."""
    base: int
    off: int
    shape: tuple
    dtype: str = "f32"
    name: str = ""

    @property
    def numel(self) -> int:
        n = 1
        for s in self.shape:
            n *= s
        return n

    @property
    def nbytes(self) -> int:
        return self.numel * ITEMSIZE[self.dtype]

    @property
    def ref(self) -> int:
        return (self.base << 56) | self.off

    def at(self, elem_offset: int, shape: tuple | None = None) -> "TRef":
        return replace(self, off=self.off + elem_offset * ITEMSIZE[self.dtype],
                       shape=self.shape if shape is None else shape)


NULL = -1


class Arena:
    """Bump allocator over the workspace base; nothing is ever freed inside a plan (288 GB HBM:
    a b5 bs-32 256x256 training plan needs ~15 GB)."""

    def __init__(self, base: int = D.BASE["WS"]):
        self.base = base
        self.top = 0
        self.names: list[tuple[str, int, int]] = []

    def alloc(self, name: str, shape: tuple, dtype: str = "f32") -> TRef:
        t = TRef(self.base, self.top, tuple(int(s) for s in shape), dtype, name)
        self.names.append((name, self.top, t.nbytes))
        self.top += (t.nbytes + ALIGN - 1) // ALIGN * ALIGN
        return t

    def mark(self) -> int:
        return self.top


class Program:
    def __init__(self, name: str = ""):
        self.name = name
        self.ops: list[tuple[str, dict]] = []

    def add(self, kind: str, **fields) -> None:
        """`_flags` (optional): S2kOp.flags bits (opdefs.FLAG_*)."""
        t, n, d, f = D.OPS[kind]
        known = set(t) | set(n) | set(d) | set(f) | {"_flags"}
        bad = set(fields) - known
        if bad:
            raise KeyError(f"{kind}: unknown fields {sorted(bad)}")
        self.ops.append((kind, fields))

    def __len__(self) -> int:
        return len(self.ops)

    def pack(self) -> np.ndarray:
        arr = np.zeros(len(self.ops), dtype=OP_DTYPE)
        arr["t"][:] = NULL
        for i, (kind, fields) in enumerate(self.ops):
            arr[i]["kind"] = D.KIND[kind]
            for k, v in fields.items():
                if k == "_flags":
                    arr[i]["flags"] = int(v)
                    continue
                a, j = D.slot(kind, k)
                if a == "t":
                    arr[i]["t"][j] = NULL if v is None else (v.ref if isinstance(v, TRef) else int(v))
                else:
                    arr[i][a][j] = v
        return arr

    def segments(self, marks: list[int]) -> list[tuple[int, int]]:
        """[begin, end) op ranges split at `marks` (used to overlap RCCL with backward)."""
        cuts = [0] + sorted(m for m in marks if 0 < m < len(self.ops)) + [len(self.ops)]
        return [(a, b) for a, b in zip(cuts[:-1], cuts[1:]) if b > a]

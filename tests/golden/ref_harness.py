"""Import the reference's modules (build container only — /root/reference never travels).

Handles the two import hazards recorded in SURVEY.md §8(c):
  1. `src/utils.py` imports `src.modules.prithvi`, which imports `timm` (not installed) — so
     in-memory `sys.modules` entries for timm are created first, backed by the oracle's
     restatement of `Block` (parity unpinned at that boundary, see oracle/vit_block_ref.py).
  2. importing `utils` opens a log file under ROOT_DIR/logs — redirect LOG_DIR to /tmp and never
     write bytecode into the read-only tree.
Used only by make_golden.py and by the optional `test_oracle_vs_reference_live` tests, which skip
when /root/reference is absent.
"""
from __future__ import annotations

import contextlib
import os
import sys
import types
from pathlib import Path

REF = Path("/root/reference")
REPO = Path(__file__).resolve().parents[2]


def available() -> bool:
    return (REF / "src" / "modules" / "efficientnet_unet.py").exists()


_loaded: dict = {}


def load():
    """Returns a namespace with the reference modules: .unet, .losses, .prithvi, .pseg, .utils."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError("/root/reference not present")
    sys.dont_write_bytecode = True
    if str(REPO) not in sys.path:
        sys.path.insert(0, str(REPO))
    from oracle import vit_block_ref

    timm = types.ModuleType("timm")
    timm_models = types.ModuleType("timm.models")
    timm_layers = types.ModuleType("timm.models.layers")
    timm_vit = types.ModuleType("timm.models.vision_transformer")
    timm_layers.to_2tuple = vit_block_ref.to_2tuple
    timm_vit.Block = vit_block_ref.Block
    timm.models = timm_models
    timm_models.layers = timm_layers
    timm_models.vision_transformer = timm_vit
    sys.modules.setdefault("timm", timm)
    sys.modules.setdefault("timm.models", timm_models)
    sys.modules.setdefault("timm.models.layers", timm_layers)
    sys.modules.setdefault("timm.models.vision_transformer", timm_vit)

    for p in (str(REF), str(REF / "src")):
        if p not in sys.path:
            sys.path.append(p)
    import src.configs.paths as paths  # noqa

    tmp = Path(os.environ.get("TMPDIR", "/tmp")) / "s2lc_ref_logs"
    tmp.mkdir(parents=True, exist_ok=True)
    paths.LOG_DIR = tmp
    import configs.paths as paths2  # the bare-import twin of the same file

    paths2.LOG_DIR = tmp

    import src.utils as ref_utils  # noqa  (side effect: logger under the redirected LOG_DIR)
    import utils as ref_utils_bare  # noqa
    import src.modules.efficientnet_unet as unet
    import src.modules.prithvi as prithvi
    import src.modules.prithvi_segmentation as pseg
    import src.losses as losses

    ns = types.SimpleNamespace(unet=unet, prithvi=prithvi, pseg=pseg, losses=losses, utils=ref_utils)
    _loaded["ns"] = ns
    return ns


@contextlib.contextmanager
def injected_rand(values):
    """Make `torch.rand` return successive entries of `values` (list of tensors) — the
    reference draws drop-connect / masking noise from the global RNG (efficientnet_unet.py:395,
    prithvi.py:267)."""
    import torch

    it = iter(values)
    orig = torch.rand

    def fake(*size, **kw):
        v = next(it)
        shape = size[0] if len(size) == 1 and isinstance(size[0], (list, tuple, torch.Size)) else size
        return v.reshape(tuple(shape)).clone()

    torch.rand = fake
    try:
        yield
    finally:
        torch.rand = orig

"""Host-side helpers on the hot path's boundary (reference: /root/reference/src/utils.py)."""
from __future__ import annotations

import torch
from torch import nn


def initialize_classification_layer_bias(layer: nn.Linear | nn.Conv2d, class_distribution: list[float]) -> None:
    """Classifier bias = log class prior (reference utils.py:174-188); 2 classes: log(p1/p0) fill."""
    dist = torch.tensor(class_distribution, dtype=torch.float32) + 1e-6
    assert torch.isclose(dist.sum(), torch.tensor(1.0)), f"Must sum to 1, got distribution: {class_distribution}"
    assert len(class_distribution) > 1, "Class distribution must have at least 2 classes"
    with torch.no_grad():
        if len(dist) == 2:
            layer.bias.fill_((dist[1] / dist[0]).log())
        else:
            layer.bias.copy_(dist.log())


# ---- Prithvi loaders (reference utils.py:52-96) -------------------------------------------------------------
def _prithvi_model_args(num_frames: int) -> dict:
    from pathlib import Path

    import yaml

    with (Path(__file__).resolve().parent / "configs" / "prithvi_config.yaml").open("r") as f:
        args = dict(yaml.safe_load(f)["model_args"])
    args["num_frames"] = num_frames
    return args


def load_untrained_prithvi(num_frames: int, _flat: bool = True):
    from .modules.prithvi import MaskedAutoencoderViT

    return MaskedAutoencoderViT(**_prithvi_model_args(num_frames), _flat=_flat)


def load_prithvi(num_frames: int, no_decoder: bool = True, weights: str | None = None, _flat: bool = True):
    """Pre-trained Prithvi-100M: pos tables are dropped from the checkpoint and re-initialised, the decoder is removed
    when `no_decoder` (reference utils.py:62-96).  `weights` defaults to $S2LC_PRITHVI_WEIGHTS or weights/Prithvi_100M.pt."""
    import os
    from pathlib import Path

    from .modules.prithvi import MaskedAutoencoderViT

    path = Path(weights or os.environ.get("S2LC_PRITHVI_WEIGHTS", "weights/Prithvi_100M.pt"))
    if not path.exists():
        raise FileNotFoundError(f"{path}: the Prithvi-100M checkpoint is not part of this repository "
                                "(use load_untrained_prithvi for random initialisation)")
    model = MaskedAutoencoderViT(**_prithvi_model_args(num_frames), _decoder=not no_decoder, _flat=False)
    state = torch.load(path, map_location="cpu", weights_only=True)      # a plain state dict of tensors: no pickle execution
    drop = ["pos_embed", "decoder_pos_embed"]
    if no_decoder:
        drop += ["decoder_embed", "mask_token", "decoder_blocks", "decoder_norm", "decoder_pred"]
    state = {k: v for k, v in state.items() if not any(k.startswith(p) for p in drop)}
    model.load_state_dict(state, strict=False)
    model.reinitialize_pos_embed()
    if _flat:
        from .plan.vit_plan import mae_layout

        model._init_flat(mae_layout(model.spec))
    return model


# ---- checkpoints written by the reference's trainers -------------------------------------------------------------
_CKPT_PREFIXES = ("net._orig_mod.", "net.", "_orig_mod.")


def strip_trainer_prefix(state: dict) -> dict:
    """Keys of a Lightning checkpoint of the reference's modules: `self.net` is the model, and when `torch.compile` is on
    (train_segmentation.py:70-75, train_mae_prithvi.py:59-64) the compiled wrapper adds `_orig_mod.`: `net._orig_mod.<name>`.
    Returns the state with the model's own names; entries of other sub-modules (loss weights, metrics) are dropped."""
    for pre in _CKPT_PREFIXES:
        if any(k.startswith(pre) for k in state):
            return {k[len(pre):]: v for k, v in state.items() if k.startswith(pre)}
    return dict(state)


def load_reference_checkpoint(model: nn.Module, path, strict: bool = True, trust_pickle: bool = False):
    """Load a checkpoint saved by the reference (`ModelCheckpoint`, train_segmentation.py:247-255 — a dict with
    `state_dict`) or a bare state dict into one of this package's modules; accepts the `net._orig_mod.` / `net.` prefixes.
    The flat parameter buffer is kept (values are copied into the existing views).  Files are read with `weights_only=True`;
    `trust_pickle=True` allows full unpickling (arbitrary code execution: only for files you trust)."""
    if isinstance(path, dict):
        ck = path
    else:
        try:        # tensors / containers only: no arbitrary pickle execution
            ck = torch.load(path, map_location="cpu", weights_only=True)
        except Exception as e:  # noqa: BLE001
            if not trust_pickle:
                raise RuntimeError(f"{path}: the checkpoint holds objects beyond tensors and plain containers (a Lightning checkpoint "
                                   "pickles its hyper-parameters); pass trust_pickle=True to unpickle a file you trust") from e
            ck = torch.load(path, map_location="cpu", weights_only=False)
    state = ck["state_dict"] if isinstance(ck, dict) and "state_dict" in ck else ck
    return model.load_state_dict(strip_trainer_prefix(state), strict=strict)

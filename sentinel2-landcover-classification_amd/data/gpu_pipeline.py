"""GPU-side input pipeline: the per-sample transform chain of the reference's loaders as ONE stage.

What it replaces (all on the CPU, per sample, one DataLoader worker in the reference's configs):
  * `S2OSMDataset.__getitem__` (src/data/s2osm_dataset.py:51-71): CNES label remap with `np.vectorize`
    (src/configs/cnes_labell_mappings.py:78-95), `c h w -> h w c` and back, `.float()`, `.long()`, `unsqueeze(1)` unless
    `squeeze_time_dim`;
  * the albumentations `Compose` of `S2OSMDatamodule.setup` (src/data/s2osm_datamodule.py:75-87) and `MAEDatamodule.setup`
    (src/data/mae_datamodule.py:60-73): RandomCrop | CenterCrop -> HorizontalFlip(p) -> VerticalFlip(p) -> Normalize(mean, std)
    (albumentations' Normalize scales mean and std by max_pixel_value = 255 — the reference passes the defaults).
Decoding the GeoTIFFs (rasterio) stays on the host: the decoded int16 tiles (12.4 k tiles x 6 x 512 x 512 x 2 B = 39 GB for
the reference's largest area) are uploaded once and stay resident in HBM; every step then costs one TILE_PREP launch.

Randomness: albumentations draws from Python's `random`; here crop offsets and flips are drawn from a `torch.Generator`
on the host with the same formulas (`int((H - S + 1) * u)`, `u < p`) and passed to the stage, like every other random
draw of this library (drop-connect, MAE masking).  Given the same draws the output is bit-identical to the reference's.
"""
from __future__ import annotations

import typing

import numpy as np
import torch

from .. import _lib
from ..plan import opdefs as D
from ..plan.program import Program, TRef

# CNES land-cover classes 1..23 grouped as in src/configs/cnes_labell_mappings.py:47-75 (class 0 = outside France)
_CNES_GROUP = {**{k: "impervious_surface" for k in (1, 2, 3, 4)},
               **{k: "agriculture" for k in (5, 6, 7, 8, 9, 10, 11, 12, 14, 15)},
               **{k: "nature" for k in (13, 16, 17, 18, 19, 20, 21, 22, 23)}}
CNES_LABEL_MAPS = {   # key order = class index (cnes_labell_mappings.py:42-45)
    "cnes-multiclass": ["other", "agriculture", "nature", "impervious_surface"],
    "cnes-impervious-binary": ["other", "impervious_surface"],
    "cnes-nature-binary": ["other", "nature"],
    "cnes-agriculture-binary": ["other", "agriculture"],
}


def label_lut(label_map_name: str) -> torch.Tensor:
    """256-entry int32 table equal to `get_cnes_transform(label_map_name, ...)` applied to every uint8 value
    (cnes_labell_mappings.py:78-95): identity unless the map is a simplified CNES one; there, label 0, labels without a
    group and groups missing from the map go to 0, the rest to the group's position in the map."""
    lut = torch.arange(256, dtype=torch.int32)
    if "cnes" in label_map_name and label_map_name != "cnes-full":
        keys = CNES_LABEL_MAPS[label_map_name]
        for v in range(256):
            g = _CNES_GROUP.get(v, "_")
            lut[v] = 0 if (v == 0 or g not in keys) else keys.index(g)
    return lut


class S2OSMSample(typing.NamedTuple):
    x: torch.Tensor
    y: torch.Tensor


class GpuTilePipeline:
    def __init__(self, mean, std, random_crop_size: int = 224, augment: bool = False, random_horizontal_flip_p: float = 0.0,
                 random_vertical_flip_p: float = 0.0, label_map: str = "osm-multiclass", squeeze_time_dim: bool = False,
                 n_time_frames: int = 1, max_pixel_value: float = 255.0, device="cuda"):
        self.S, self.augment = int(random_crop_size), bool(augment)
        self.hp, self.vp = float(random_horizontal_flip_p), float(random_vertical_flip_p)
        self.squeeze, self.frames = bool(squeeze_time_dim), int(n_time_frames)
        self.device = torch.device(device)
        # albumentations.normalize: mean, std as float32 scaled by max_pixel_value (float32 products), float32 reciprocal
        m = np.array(torch.as_tensor(mean).tolist(), dtype=np.float32)
        s = np.array(torch.as_tensor(std).tolist(), dtype=np.float32)
        m *= np.float32(max_pixel_value)
        s *= np.float32(max_pixel_value)
        self.norm = torch.from_numpy(np.stack([m, np.reciprocal(s, dtype=np.float32)])).to(self.device)
        self.lut = label_lut(label_map).to(self.device)
        self.raw = self.labels = None
        if self.S % 4:
            raise ValueError("random_crop_size must be a multiple of 4")

    def load(self, raw: torch.Tensor, labels: torch.Tensor | None) -> None:
        """raw: int16 [N, C, H, W] decoded tiles; labels: uint8 [N, H, W] (None for the MAE loaders).  Kept resident."""
        if raw.dtype != torch.int16 or raw.dim() != 4:
            raise ValueError("raw tiles must be int16 [N, C, H, W]")
        if labels is not None and (labels.dtype != torch.uint8 or labels.shape != (raw.shape[0],) + tuple(raw.shape[2:])):
            raise ValueError("labels must be uint8 [N, H, W]")
        if raw.shape[1] != self.norm.shape[1]:
            raise ValueError("mean / std length differs from the number of bands")
        if self.S > raw.shape[2] or self.S > raw.shape[3]:
            raise ValueError("crop larger than the tile")   # albumentations raises ValueError here too
        self.raw = raw.to(self.device).contiguous()
        self.labels = None if labels is None else labels.to(self.device).contiguous()

    def draw_params(self, indices, training: bool, generator: torch.Generator | None = None) -> torch.Tensor:
        """int32 [B, 4] = {tile, y0, x0, flips}: albumentations' coordinate formulas on host-drawn uniforms."""
        idx = torch.as_tensor(indices, dtype=torch.int64).reshape(-1)
        B, (H, W), S = idx.numel(), self.raw.shape[2:], self.S
        par = torch.zeros(B, 4, dtype=torch.int32)
        par[:, 0] = idx.to(torch.int32)
        if training and self.augment:
            u = torch.rand(B, 4, generator=generator, dtype=torch.float64)
            par[:, 1] = ((H - S + 1) * u[:, 0]).to(torch.int32)        # get_random_crop_coords: int((height - crop + 1) * h_start)
            par[:, 2] = ((W - S + 1) * u[:, 1]).to(torch.int32)
            par[:, 3] = (u[:, 2] < self.hp).to(torch.int32) + 2 * (u[:, 3] < self.vp).to(torch.int32)
        else:
            par[:, 1], par[:, 2] = (H - S) // 2, (W - S) // 2            # get_center_crop_coords
        return par

    @torch.no_grad()
    def __call__(self, indices=None, training: bool = True, params: torch.Tensor | None = None,
                 generator: torch.Generator | None = None) -> S2OSMSample:
        if self.raw is None:
            raise RuntimeError("call load() first")
        if not self.raw.is_cuda:
            raise RuntimeError("GpuTilePipeline runs on the GPU (there is no CPU fallback)")
        par = self.draw_params(indices, training, generator) if params is None else torch.as_tensor(params, dtype=torch.int32).cpu()
        N, C, H, W = self.raw.shape
        S, B = self.S, par.shape[0]
        ok = (par[:, 0] >= 0) & (par[:, 0] < N) & (par[:, 1] >= 0) & (par[:, 1] <= H - S) & (par[:, 2] >= 0) & (par[:, 2] <= W - S) \
            & (par[:, 3] >= 0) & (par[:, 3] <= 3)
        if par.shape[1:] != (4,) or not bool(ok.all()):
            raise ValueError("params out of range (tile index, crop offsets, flip bits)")
        par_d = par.contiguous().to(self.device)
        x = torch.empty(B, C, S, S, dtype=torch.float32, device=self.device)
        y = torch.empty(B, S, S, dtype=torch.int64, device=self.device) if self.labels is not None else None
        prog = Program()
        prog.add("TILE_PREP", RAW=TRef(D.BASE["X"], 0, (N, C, H, W), "i16"),
                 LABELS=TRef(D.BASE["Y"], 0, (N, H, W), "u8") if y is not None else None,
                 PARAMS=TRef(D.BASE["AUX"], 0, (B, 4), "i32"), NORM=TRef(D.BASE["CONST"], 0, (2, C)),
                 LUT=TRef(D.BASE["NOISE"], 0, (256,), "i32") if y is not None else None,
                 X=TRef(D.BASE["OUT"], 0, (B, C, S, S)), Y=TRef(D.BASE["DOUT"], 0, (B, S, S), "i64") if y is not None else None,
                 B=B, C=C, H=H, W=W, S=S, NSRC=N)
        bases = _lib.Bases().set("X", self.raw).set("AUX", par_d).set("CONST", self.norm).set("OUT", x)
        if y is not None:
            bases.set("Y", self.labels).set("NOISE", self.lut).set("DOUT", y)
        _lib.run(prog.pack(), bases, torch.cuda.current_stream(self.device).cuda_stream)
        if not self.squeeze and self.frames == 1:
            x = x.unsqueeze(2)            # per sample (c, 1, h, w), s2osm_dataset.py:64-66
        return S2OSMSample(x=x, y=y)

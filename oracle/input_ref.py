"""TEST INFRASTRUCTURE ONLY — CPU restatement (numpy) of the reference's per-sample input chain, the oracle for the
TILE_PREP stage and `s2lc_amd.data.gpu_pipeline`.  Only tests/ may import this.

Follows, function by function:
  cnes_transform        /root/reference/src/configs/cnes_labell_mappings.py:78-95 (get_cnes_transform, _cnes_transform)
  sample_transform      /root/reference/src/data/s2osm_dataset.py:51-71 (__getitem__) with the Compose built at
                        /root/reference/src/data/s2osm_datamodule.py:75-87
albumentations (requirements.txt: `albumentations>=1.3.1`, a floor, no lock file, not installed here) is third party: its
published 1.3.1 algorithms are restated below — `functional.normalize` (numpy path, taken for != 3 channels),
`crops.functional.get_center_crop_coords`, `get_random_crop_coords`, `hflip` / `vflip`.  PARITY UNPINNED at that boundary
(no reference fixture exercises it); the reference's own code above it is restated line by line.
"""
from __future__ import annotations

import numpy as np

CNES_TO_GROUP = {1: "impervious_surface", 2: "impervious_surface", 3: "impervious_surface", 4: "impervious_surface",
                 5: "agriculture", 6: "agriculture", 7: "agriculture", 8: "agriculture", 9: "agriculture", 10: "agriculture",
                 11: "agriculture", 12: "agriculture", 13: "nature", 14: "agriculture", 15: "agriculture", 16: "nature",
                 17: "nature", 18: "nature", 19: "nature", 20: "nature", 21: "nature", 22: "nature", 23: "nature"}


def cnes_transform(labels: np.ndarray, label_map_name: str, label_map_keys: list[str]) -> np.ndarray:
    if not ("cnes" in label_map_name and label_map_name != "cnes-full"):
        return labels
    out = np.empty(labels.shape, dtype=int)
    flat_in, flat_out = labels.reshape(-1), out.reshape(-1)
    for i, label in enumerate(flat_in):            # np.vectorize(map_func) in the reference
        target = CNES_TO_GROUP.get(int(label), "_")
        flat_out[i] = 0 if (label == 0 or target not in label_map_keys) else label_map_keys.index(target)
    return out


def center_crop_coords(height: int, width: int, crop: int):
    y1, x1 = (height - crop) // 2, (width - crop) // 2
    return y1, x1


def random_crop_coords(height: int, width: int, crop: int, h_start: float, w_start: float):
    return int((height - crop + 1) * h_start), int((width - crop + 1) * w_start)


def normalize(img: np.ndarray, mean, std, max_pixel_value: float = 255.0) -> np.ndarray:
    mean = np.array(mean, dtype=np.float32)
    mean *= max_pixel_value
    std = np.array(std, dtype=np.float32)
    std *= max_pixel_value
    denominator = np.reciprocal(std, dtype=np.float32)
    img = img.astype(np.float32)
    img -= mean
    img *= denominator
    return img


def sample_transform(sentinel: np.ndarray, osm: np.ndarray | None, y1: int, x1: int, crop: int, hflip: bool, vflip: bool, mean, std,
                     squeeze_time_dim: bool, n_time_frames: int = 1):
    """sentinel int16 [C,H,W], osm [H,W] (already through cnes_transform) -> (x float32, y int64) as __getitem__ returns them"""
    img = np.transpose(sentinel, (1, 2, 0))                       # "c h w -> h w c"
    img = img[y1:y1 + crop, x1:x1 + crop]
    msk = None if osm is None else osm[y1:y1 + crop, x1:x1 + crop]
    if hflip:
        img = np.ascontiguousarray(img[:, ::-1, ...])
        msk = None if msk is None else np.ascontiguousarray(msk[:, ::-1, ...])
    if vflip:
        img = np.ascontiguousarray(img[::-1, ...])
        msk = None if msk is None else np.ascontiguousarray(msk[::-1, ...])
    img = normalize(img, mean, std)
    x = np.transpose(img, (2, 0, 1)).astype(np.float32)           # "h w c -> c h w", .float()
    if not squeeze_time_dim and n_time_frames == 1:
        x = x[:, None]
    return x, (None if msk is None else msk.astype(np.int64))

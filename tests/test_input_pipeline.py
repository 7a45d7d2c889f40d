"""Input pipeline: TILE_PREP against oracle/ops_ref.py and the whole `GpuTilePipeline` against the
numpy restatement of the reference's per-sample chain (oracle/input_ref.py).  Integer outputs and the normalised floats
must be BIT-EXACT (the normalisation is two separately rounded fp32 operations in numpy and in the kernel)."""
import numpy as np
import pytest
import torch

import s2lc_amd  # noqa: F401
from oracle import input_ref as R
from s2lc_amd.data.gpu_pipeline import CNES_LABEL_MAPS, GpuTilePipeline, label_lut

MEAN = [0.045, 0.0612, 0.0433, 0.2281, 0.1503, 0.0911]      # of the same magnitude as prithvi_config.yaml's, / 255-scaled
STD = [0.0210, 0.0183, 0.0267, 0.0551, 0.0492, 0.0405]


def _tiles(n, c, h, w, seed):
    g = torch.Generator().manual_seed(seed)
    raw = torch.randint(-200, 9000, (n, c, h, w), generator=g, dtype=torch.int32).to(torch.int16)
    lab = torch.randint(0, 30, (n, h, w), generator=g, dtype=torch.int32).to(torch.uint8)
    return raw, lab


@pytest.mark.parametrize("name", ["cnes-multiclass", "cnes-impervious-binary", "cnes-nature-binary", "cnes-agriculture-binary",
                                  "cnes-full", "osm-multiclass"])
def test_label_lut_equals_the_vectorised_remap(name):
    vals = np.arange(256, dtype=np.uint8).reshape(16, 16)
    want = R.cnes_transform(vals, name, CNES_LABEL_MAPS.get(name, []))
    assert np.array_equal(label_lut(name).numpy().reshape(16, 16), want)


def test_crop_coordinates_follow_albumentations():
    p = GpuTilePipeline(MEAN, STD, random_crop_size=224, augment=True, random_horizontal_flip_p=0.5, random_vertical_flip_p=0.5, device="cpu")
    p.raw = torch.zeros(3, 6, 512, 512, dtype=torch.int16)
    assert p.draw_params([0, 2], training=False)[:, 1:].tolist() == [[144, 144, 0], [144, 144, 0]]
    g = torch.Generator().manual_seed(3)
    par = p.draw_params([1] * 64, training=True, generator=g)
    u = torch.rand(64, 4, generator=torch.Generator().manual_seed(3), dtype=torch.float64)
    for b in range(64):
        assert (int(par[b, 1]), int(par[b, 2])) == R.random_crop_coords(512, 512, 224, float(u[b, 0]), float(u[b, 1]))
        assert int(par[b, 3]) == int(u[b, 2] < 0.5) + 2 * int(u[b, 3] < 0.5)
    assert par[:, 1].max() <= 288 and par[:, 3].max() <= 3


def test_pipeline_refuses_cpu():
    p = GpuTilePipeline(MEAN, STD, device="cpu")
    p.load(*_tiles(1, 6, 256, 256, 0))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        p([0], training=False)


@pytest.mark.gpu
@pytest.mark.parametrize("C,H,W,S,B,with_labels", [(6, 64, 80, 32, 5, True), (13, 40, 40, 40, 2, True), (6, 512, 512, 224, 3, False)])
def test_tile_prep_stage_bit_exact(C, H, W, S, B, with_labels):
    from s2lc_amd.plan import opdefs as D  # noqa: F401
    from tests.test_ops_gpu import Case

    c = Case(31)
    n = 3
    rawd, labd = _tiles(n, C, H, W, 5)
    raw = c.t("raw", (n, C, H, W), rawd, "i16")
    lab = c.t("lab", (n, H, W), labd, "u8") if with_labels else None
    g = c.gen
    par = torch.stack([torch.randint(0, n, (B,), generator=g), torch.randint(0, H - S + 1, (B,), generator=g),
                       torch.randint(0, W - S + 1, (B,), generator=g), torch.arange(B) % 4], 1)
    params = c.t("params", (B, 4), par, "i32")
    norm = c.t("norm", (2, C), torch.stack([torch.rand(C, generator=g) * 3000, torch.rand(C, generator=g) * 1e-3 + 1e-4]))
    lut = c.t("lut", (256,), torch.randint(0, 5, (256,), generator=g), "i32") if with_labels else None
    x = c.t("x", (B, C, S, S), "nan")
    y = c.t("y", (B, S, S), torch.full((B, S, S), -7), "i64") if with_labels else None
    c.run("TILE_PREP", ["x"] + (["y"] if with_labels else []), tol=1e-30, RAW=raw, LABELS=lab, PARAMS=params, NORM=norm, LUT=lut, X=x, Y=y,
          B=B, C=C, H=H, W=W, S=S, NSRC=n)


@pytest.mark.gpu
@pytest.mark.parametrize("label_map,squeeze,augment", [("cnes-multiclass", True, True), ("osm-multiclass", False, True),
                                                       ("cnes-nature-binary", True, False)])
def test_pipeline_matches_reference_chain_bit_exact(label_map, squeeze, augment):
    n, C, H, W, S = 4, 6, 96, 128, 64
    raw, lab = _tiles(n, C, H, W, 11)
    pipe = GpuTilePipeline(MEAN, STD, random_crop_size=S, augment=augment, random_horizontal_flip_p=0.5, random_vertical_flip_p=0.5,
                           label_map=label_map, squeeze_time_dim=squeeze)
    pipe.load(raw, lab)
    idx = [3, 0, 2, 2, 1, 3]
    par = pipe.draw_params(idx, training=True, generator=torch.Generator().manual_seed(8))
    out = pipe(params=par)
    assert out.x.shape == ((len(idx), C, S, S) if squeeze else (len(idx), C, 1, S, S)) and out.y.dtype == torch.int64
    keys = CNES_LABEL_MAPS.get(label_map, [])
    for b, (src, y0, x0, fl) in enumerate(par.tolist()):
        osm = R.cnes_transform(lab[src].numpy(), label_map, keys)
        wx, wy = R.sample_transform(raw[src].numpy(), osm, y0, x0, S, bool(fl & 1), bool(fl & 2), MEAN, STD, squeeze)
        assert np.array_equal(out.x[b].cpu().numpy(), wx), f"sample {b}: normalised crop differs"
        assert np.array_equal(out.y[b].cpu().numpy(), wy), f"sample {b}: labels differ"
    if not augment:
        assert par[:, 1:].tolist() == [[(H - S) // 2, (W - S) // 2, 0]] * len(idx)


@pytest.mark.gpu
def test_pipeline_rejects_out_of_range_params():
    pipe = GpuTilePipeline(MEAN, STD, random_crop_size=32)
    pipe.load(*_tiles(2, 6, 64, 64, 1))
    for bad in ([[2, 0, 0, 0]], [[0, 33, 0, 0]], [[0, 0, -1, 0]], [[0, 0, 0, 4]]):
        with pytest.raises(ValueError):
            pipe(params=torch.tensor(bad, dtype=torch.int32))

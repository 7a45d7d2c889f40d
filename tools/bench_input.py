"""Throughput of the GPU input pipeline (TILE_PREP): 6 x 512 x 512 int16 tiles -> normalised 224 crops + remapped labels."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import s2lc_amd  # noqa: E402,F401
from s2lc_amd.data.gpu_pipeline import GpuTilePipeline  # noqa: E402


def main():
    n, C, H, S, B = 256, 6, 512, 224, 256
    raw = torch.randint(0, 9000, (n, C, H, H), dtype=torch.int32).to(torch.int16)
    lab = torch.randint(0, 24, (n, H, H), dtype=torch.int32).to(torch.uint8)
    pipe = GpuTilePipeline([0.05] * C, [0.02] * C, random_crop_size=S, augment=True, random_horizontal_flip_p=0.5,
                           random_vertical_flip_p=0.5, label_map="cnes-multiclass", squeeze_time_dim=True)
    pipe.load(raw, lab)
    g = torch.Generator().manual_seed(0)
    par = pipe.draw_params(torch.randint(0, n, (B,), generator=g), training=True, generator=g)
    for _ in range(3):
        pipe(params=par)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    iters = 20
    for _ in range(iters):
        pipe(params=par)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    byts = B * S * S * (C * (2 + 4) + 1 + 8)
    print(f"TILE_PREP B={B}: {dt * 1e3:.3f} ms/batch  {B / dt:.0f} tiles/s  {byts / dt / 1e12:.2f} TB/s (host param draw + upload included)")


if __name__ == "__main__":
    main()

"""Capture one U-Net training step (forward + focal loss + backward) in a HIP graph and compare replay with eager launches."""
import sys
import time
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import s2lc_amd  # noqa: E402,F401
from s2lc_amd.losses import FocalLoss  # noqa: E402
from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet  # noqa: E402


def main():
    dev = torch.device("cuda:0")
    B, C, H = 32, 13, 256
    model = EfficientnetUnet(EfficientNetConfig("b5", C, 4, class_distribution=[.25] * 4)).to(dev).train()
    loss_fn = FocalLoss(torch.ones(4), 2.0, 0.0, ignore_index=0)
    x = torch.randn(B, C, H, H, device=dev)
    y = torch.randint(0, 4, (B, H, H), device=dev)
    model.drop_connect_noise = torch.rand(39, B, device=dev)

    def step():
        for p in model.parameters():
            p.grad = None
        loss = loss_fn(model(x), y)
        loss.backward()
        return loss

    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    eager = (time.perf_counter() - t0) / 10
    g_eager = model._grad_buffer().clone()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        loss = step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        graph.replay()
    torch.cuda.synchronize()
    rep = (time.perf_counter() - t0) / 10
    err = (model._grad_buffer() - g_eager).abs().max().item() / g_eager.abs().max().item()
    print(f"eager {eager * 1e3:.2f} ms/step, graph replay {rep * 1e3:.2f} ms/step, loss {loss.item():.5f}, grad rel diff {err:.2e}")


if __name__ == "__main__":
    main()

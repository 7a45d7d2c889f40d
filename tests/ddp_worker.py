"""One rank of the data-parallel rehearsal (tests/test_ddp_gpu.py): a fresh process per rank, all ranks on cuda:0,
gradients reduced over gloo by the product's FlatGradReducer while the segmented backward runs.

    python tests/ddp_worker.py RANK WORLD PORT OUTDIR CASE

Mirrors what Lightning's implicit DDP does around the reference's training_step
(/root/reference/src/train_segmentation.py:273-280): identical start weights (broadcast from rank 0), a shard of the
global batch per rank, per-rank mean loss, gradients averaged over ranks, BatchNorm statistics per rank."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def build_case(case: str, seed: int):
    """(model on the CPU, global batch x, labels y or None, per-sample noise or None, loss function)."""
    import torch

    import s2lc_amd  # noqa: F401
    from oracle import detgen
    from oracle import efficientnet_unet_ref as R

    if case.startswith("unet"):
        from s2lc_amd.losses import FocalLoss
        from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet

        net = R.build("b0", 4, 4, drop_connect_rate=0.2)
        sd = detgen.fill_state(R.state_shapes(net), seed=seed)
        model = EfficientnetUnet(EfficientNetConfig("b0", 4, 4, class_distribution=[0.25] * 4, drop_connect_rate=0.2))
        model.load_state_dict(sd)
        model._bucket_floats = 1 << 20          # several gradient buckets even for a b0
        x = detgen.normal("ddp.x", (4, 4, 64, 64), seed=77)
        y = detgen.labels("ddp.y", (4, 64, 64), 4, seed=77)
        noise = detgen.uniform("ddp.dc", (len(net.blocks), 4), 0.0, 1.0, seed=77)
        loss_fn = FocalLoss(torch.ones(4), 2.0, 0.0, ignore_index=0)
        return model, x, y, noise, loss_fn
    from s2lc_amd.modules.prithvi import MaskedAutoencoderViT
    from tests.helpers import PRITHVI_SMALL

    torch.manual_seed(seed)
    model = MaskedAutoencoderViT(**PRITHVI_SMALL)
    for p in model.parameters():
        if p.requires_grad:
            p.data.copy_(detgen.normal(f"ddp.mae.{seed}.{tuple(p.shape)}", tuple(p.shape), seed=seed) * 0.05)
    model._bucket_floats = 4096
    x = detgen.normal("ddp.mae.x", (4, 3, 1, 32, 32), seed=78)
    noise = detgen.uniform("ddp.mae.noise", (4, 16), 0.0, 1.0, seed=78)
    return model, x, None, noise, None


def run_shard(model, case, x, y, noise, loss_fn, lo, hi, dev):
    """forward + backward of samples [lo, hi) of the global batch; returns the loss."""
    xs = x[lo:hi].to(dev)
    case = case.replace("_accum", "").replace("_sharded", "").replace("_adam", "")
    if case.startswith("unet"):
        model.train(case == "unet_train")
        model.drop_connect_noise = noise[:, lo:hi].contiguous() if case == "unet_train" else None
        loss = loss_fn(model(xs), y[lo:hi].to(dev))
    elif case == "mae_methods":
        # the reference's forward written out by a caller (prithvi.py:352-356): three autograd nodes in one backward
        model.masking_noise = noise[lo:hi].contiguous()
        latent, mask, ids = model.forward_encoder(xs, 0.75)
        loss = model.forward_loss(xs, model.forward_decoder(latent, ids), mask)
    else:
        model.masking_noise = noise[lo:hi].contiguous()
        loss, _, _ = model(xs, mask_ratio=0.75)
    loss.backward()
    return loss


def main():
    rank, world, port, outdir, case = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4], sys.argv[5]
    backend = sys.argv[6] if len(sys.argv) > 6 else "gloo"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist

    dev = torch.device("cuda:0")
    if backend == "nccl":       # RCCL: one communicator per device, bound at init (what bench.py does for N > 1)
        torch.cuda.set_device(dev)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)
    try:
        from s2lc_amd.ddp import FlatGradReducer

        model, x, y, noise, loss_fn = build_case(case, seed=5 + rank)      # different weights per rank: the broadcast must fix that
        model.to(dev)
        red = FlatGradReducer(model, dist, mode="sharded" if case.endswith("_sharded") else "allreduce")
        red.broadcast_parameters(0)
        torch.cuda.synchronize()
        w0 = model._flat_params.detach().cpu().clone()
        calls = []
        inner = model._bwd_segment_hook

        def hook(lo, hi, grads):
            calls.append((lo, hi))
            inner(lo, hi, grads)

        model._bwd_segment_hook = hook
        per = x.shape[0] // world
        if case.endswith("_sharded") or case.endswith("_adam"):
            # three optimiser steps: two with bucketed reduction, the third accumulated (reduced as one range by finish(): in the
            # sharded mode the slices change owners there).  "_adam": all-reduce + full Adam on every rank; "_sharded": reduce-scatter,
            # Adam on the owned slices, all-gather of the parameters - same weights and (consolidated) moments, bit for bit
            from s2lc_amd.optim import FlatAdam

            opt = FlatAdam(model, lr=1e-2, weight_decay=0.01, reducer=red)
            half = per // 2
            for step in range(3):
                opt.zero_grad()
                if step == 2:
                    with red.no_sync():
                        run_shard(model, case, x, y, noise, loss_fn, rank * per, rank * per + half, dev)
                    loss = run_shard(model, case, x, y, noise, loss_fn, rank * per + half, (rank + 1) * per, dev)
                else:
                    loss = run_shard(model, case, x, y, noise, loss_fn, rank * per, (rank + 1) * per, dev)
                red.finish()
                opt.step()
            refused = False
            if case.endswith("_sharded"):
                try:
                    opt.state_dict()
                except RuntimeError:
                    refused = True
            opt.consolidate_state()
            sd = opt.state_dict()["state"]
            torch.cuda.synchronize()
            torch.save(dict(rank=rank, backend=dist.get_backend(), params=model._flat_params.detach().cpu(), m=sd["exp_avg"].cpu(), v=sd["exp_avg_sq"].cpu(),
                            w0=w0, loss=float(loss), calls=calls, native=red._native_rs, refused=refused, segments=list(red.last_segments),
                            grads=model._grad_buffer().detach().cpu(), bufs=model._flat_bufs.detach().cpu()), os.path.join(outdir, f"{case}.r{rank}.pt"))
            return
        if case.endswith("_accum"):
            # gradient accumulation as with torch DDP: every micro-batch but the last inside no_sync(), ONE reduction of the sum
            half = per // 2
            with red.no_sync():
                run_shard(model, case, x, y, noise, loss_fn, rank * per, rank * per + half, dev)
            loss = run_shard(model, case, x, y, noise, loss_fn, rank * per + half, (rank + 1) * per, dev)
        else:
            loss = run_shard(model, case, x, y, noise, loss_fn, rank * per, (rank + 1) * per, dev)
        red.finish()
        torch.cuda.synchronize()
        torch.save(dict(rank=rank, backend=dist.get_backend(), grads=model._grad_buffer().detach().cpu(), w0=w0, loss=float(loss), calls=calls,
                        bufs=model._flat_bufs.detach().cpu()), os.path.join(outdir, f"{case}.r{rank}.pt"))
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

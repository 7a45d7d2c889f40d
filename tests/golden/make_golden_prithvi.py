"""Prithvi fixtures, generated from the imported reference with the oracle's restated
timm Block plugged into sys.modules (parity UNPINNED for the block body, pinned for everything around it).

Run through make_golden.py:   python tests/golden/make_golden.py prithvi
"""
from __future__ import annotations

import contextlib
from pathlib import Path

import numpy as np
import torch

import ref_harness
from oracle import detgen
from oracle import prithvi_ref as P

HERE = Path(__file__).resolve().parent

SMALL = dict(img_size=32, patch_size=8, num_frames=1, tubelet_size=1, in_chans=3, embed_dim=32, depth=2, num_heads=2,
             decoder_embed_dim=16, decoder_depth=1, decoder_num_heads=2)
SMALL_T3 = dict(SMALL, num_frames=3)
SEG_SMALL = dict(SMALL, img_size=64, patch_size=16)   # the neck upsamples x16: output = image only when patch = 16
FULL = dict(P.PRITHVI_100M, num_frames=1)


def sub(t, n=2048):
    f = t.detach().reshape(-1)
    step = max(1, f.numel() // n)
    return f[::step][:n].to(torch.float32).numpy().copy()


def checks(t):
    d = t.detach().double()
    return np.array([d.sum().item(), d.abs().sum().item(), d.abs().max().item(), float(d.numel())])


def fill_mae(model, cfg: P.MaeCfg, seed: int, decoder=True, prefix=""):
    shapes = P.mae_state_shapes(cfg, decoder=decoder)
    sd = detgen.fill_state(shapes, seed=seed)
    sd["pos_embed"] = P.sincos_pos_embed(cfg.embed_dim, cfg.grid)
    sd["decoder_pos_embed"] = P.sincos_pos_embed(cfg.decoder_embed_dim, cfg.grid)
    return sd


def mae_case(ns, tag, args, B, mask_ratio, seed, grads):
    torch.manual_seed(0)
    model = ns.prithvi.MaskedAutoencoderViT(**args)
    cfg = P.MaeCfg(**args)
    sd = fill_mae(model, cfg, seed)
    # the reference's own tables must equal the oracle's restatement bit for bit
    assert torch.equal(model.pos_embed.data, sd["pos_embed"]) and torch.equal(model.decoder_pos_embed.data, sd["decoder_pos_embed"])
    res = model.load_state_dict(sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    assert list(model.state_dict().keys()) == list(sd.keys()), "state-dict order"
    x = detgen.normal(f"{tag}.x", (B, cfg.in_chans, cfg.num_frames, cfg.img_size, cfg.img_size), seed=seed)
    noise = detgen.uniform(f"{tag}.noise", (B, cfg.num_patches), 0.0, 1.0, seed=seed)
    out = {"args": np.array([str(sorted(args.items()))]), "meta": np.array([B, seed]), "mask_ratio": np.array([mask_ratio])}
    with ref_harness.injected_rand([noise]):
        loss, pred, mask = model(x, mask_ratio=mask_ratio)
    with ref_harness.injected_rand([noise]):
        latent, mask2, ids_restore = model.forward_encoder(x, mask_ratio)
    assert torch.equal(mask, mask2)
    out["loss"] = np.array([loss.item()])
    out["pred_sub"] = sub(pred, 4096)
    out["pred_ck"] = checks(pred)
    out["mask"] = mask.to(torch.uint8).numpy()
    out["ids_restore"] = ids_restore.to(torch.int32).numpy()
    out["latent_sub"] = sub(latent, 4096)
    out["latent_ck"] = checks(latent)
    if grads:
        loss.backward()
        named = dict(model.named_parameters())
        tot = 0.0
        for k, p in named.items():
            if p.grad is not None:
                tot += p.grad.double().pow(2).sum().item()
        out["grad_total_sq"] = np.array([tot])
        out["grad_none"] = np.array([k for k, p in named.items() if p.grad is None])
        for g in ["cls_token", "mask_token", "patch_embed.proj.weight", "patch_embed.proj.bias", "blocks.0.norm1.weight",
                  "blocks.0.attn.qkv.weight", "blocks.0.attn.qkv.bias", "blocks.1.attn.proj.weight", "blocks.1.mlp.fc1.weight",
                  "blocks.1.mlp.fc2.bias", "norm.bias", "decoder_embed.weight", "decoder_blocks.0.attn.qkv.weight",
                  "decoder_blocks.0.norm2.bias", "decoder_norm.weight", "decoder_pred.weight", "decoder_pred.bias"]:
            if g in named and named[g].grad is not None:
                out["grad:" + g] = sub(named[g].grad, 512)
                out["gradck:" + g] = checks(named[g].grad)
    np.savez_compressed(HERE / f"prithvi_mae_{tag}.npz", **out)
    print(f"prithvi_mae_{tag}: loss {out['loss']}, pred ck {out['pred_ck']}")


@contextlib.contextmanager
def injected_dropout2d(u_list, p_expected):
    """Dropout2d draws its per-(sample, channel) Bernoulli mask with bernoulli_ (not torch.rand): replace
    F.dropout2d by the documented semantics on an injected uniform (kept iff u >= p, scaled by 1/(1-p))."""
    import torch.nn.functional as F

    it = iter(u_list)
    orig = F.dropout2d

    def fake(inp, p=0.5, training=True, inplace=False):
        if not training or p == 0.0:
            return inp
        assert abs(p - p_expected) < 1e-12
        u = next(it)
        keep = (u >= p).to(inp.dtype) / (1.0 - p)
        return inp * keep[:, :, None, None]

    F.dropout2d = fake
    try:
        yield
    finally:
        F.dropout2d = orig


def seg_case(ns, tag, args, B, ncls, fcn_out, frozen, train, seed, grads):
    torch.manual_seed(0)
    mcfg = P.MaeCfg(**args)

    def fake_load_prithvi(num_frames, no_decoder=True):
        a = dict(args)
        a["num_frames"] = num_frames
        m = ns.prithvi.MaskedAutoencoderViT(**a)
        for attr in ["decoder_embed", "mask_token", "decoder_blocks", "decoder_norm", "decoder_pred"]:
            delattr(m, attr)  # what load_prithvi(no_decoder=True) does (utils.py:75-86); the checkpoint itself is absent
        return m

    orig = ns.pseg.load_prithvi
    ns.pseg.load_prithvi = fake_load_prithvi
    try:
        g = mcfg.img_size // mcfg.patch_size
        rc = ns.pseg.PrithviSegmentationNetConfig(num_frames=mcfg.num_frames, num_classes=ncls, fcn_out_channels=fcn_out,
                                                  fcn_num_convs=1, fcn_dropout=0.1, frozen_backbone=frozen,
                                                  embed_dim=mcfg.embed_dim, patch_height=g, patch_width=g)
        model = ns.pseg.PrithviSegmentationNet(rc)
    finally:
        ns.pseg.load_prithvi = orig
    cfg = P.SegCfg(mae=mcfg, num_classes=ncls, fcn_out_channels=fcn_out, fcn_num_convs=1, fcn_dropout=0.1, frozen_backbone=frozen)
    shapes = P.seg_state_shapes(cfg)
    sd = detgen.fill_state(shapes, seed=seed)
    sd["backbone.pos_embed"] = P.sincos_pos_embed(mcfg.embed_dim, mcfg.grid)
    sd["backbone.decoder_pos_embed"] = P.sincos_pos_embed(mcfg.decoder_embed_dim, mcfg.grid)
    assert list(model.state_dict().keys()) == list(sd.keys()), (list(model.state_dict().keys())[-20:], list(sd.keys())[-20:])
    model.load_state_dict(sd, strict=True)
    x = detgen.normal(f"{tag}.x", (B, mcfg.in_chans, mcfg.num_frames, mcfg.img_size, mcfg.img_size), seed=seed)
    y = detgen.labels(f"{tag}.y", (B, mcfg.img_size, mcfg.img_size), ncls, seed=seed)
    noise = detgen.uniform(f"{tag}.noise", (B, mcfg.num_patches), 0.0, 1.0, seed=seed)
    drop_u = detgen.uniform(f"{tag}.drop", (B, fcn_out), 0.0, 1.0, seed=seed)
    out = {"args": np.array([str(sorted(args.items()))]), "meta": np.array([B, ncls, fcn_out, int(frozen), int(train), seed])}
    if train:
        model.train()
        if frozen:
            model.backbone.eval()
        with ref_harness.injected_rand([noise]), injected_dropout2d([drop_u], 0.1):
            logits = model(x)
        loss = torch.nn.CrossEntropyLoss(ignore_index=0)(logits, y)
        out["loss_ce"] = np.array([loss.item()])
        if grads:
            loss.backward()
            named = dict(model.named_parameters())
            tot = 0.0
            for k, p in named.items():
                if p.grad is not None:
                    tot += p.grad.double().pow(2).sum().item()
            out["grad_total_sq"] = np.array([tot])
            out["grad_none"] = np.array([k for k, p in named.items() if p.grad is None])
            for gname in ["backbone.cls_token", "backbone.patch_embed.proj.weight", "backbone.blocks.0.attn.qkv.weight",
                          "backbone.blocks.1.mlp.fc2.weight", "backbone.norm.weight", "neck.feature_pyramid_net.0.weight",
                          "neck.feature_pyramid_net.1.ln.weight", "neck.feature_pyramid_net.3.bias",
                          "neck.feature_pyramid_net.4.weight", "neck.feature_pyramid_net.5.ln.bias",
                          "neck.feature_pyramid_net.7.weight", "head.net.0.weight", "head.net.0.bias", "head.net.1.weight",
                          "head.net.4.weight", "head.net.4.bias"]:
                if gname in named and named[gname].grad is not None:
                    out["grad:" + gname] = sub(named[gname].grad, 512)
                    out["gradck:" + gname] = checks(named[gname].grad)
        new_sd = model.state_dict()
        out["rm:head.net.1"] = new_sd["head.net.1.running_mean"].numpy().copy()
        out["rv:head.net.1"] = new_sd["head.net.1.running_var"].numpy().copy()
        out["nbt:head.net.1"] = np.array([int(new_sd["head.net.1.num_batches_tracked"])])
    else:
        model.eval()
        with torch.no_grad(), ref_harness.injected_rand([noise]):
            logits = model(x)
        out["loss_ce"] = np.array([torch.nn.CrossEntropyLoss(ignore_index=0)(logits, y).item()])
    out["logits_sub"] = sub(logits, 4096)
    out["logits_ck"] = checks(logits)
    out["mask"] = logits.argmax(dim=1).to(torch.uint8).numpy()
    top2 = logits.detach().topk(2, dim=1).values
    out["margin_min"] = np.array([(top2[:, 0] - top2[:, 1]).min().item()])
    np.savez_compressed(HERE / f"prithvi_seg_{tag}.npz", **out)
    print(f"prithvi_seg_{tag}: logits ck {out['logits_ck']}, ce {out['loss_ce']}")


def gen_misc(ns):
    out = {}
    for dim, grid in ((768, (1, 14, 14)), (512, (1, 14, 14)), (512, (3, 14, 14)), (32, (3, 2, 2)), (16, (1, 4, 4))):
        t = torch.from_numpy(ns.prithvi.get_3d_sincos_pos_embed(dim, grid, cls_token=True)).float()
        key = f"pos:{dim}:{grid[0]}x{grid[1]}x{grid[2]}"
        out[key + ":ck"] = checks(t)
        out[key + ":sub"] = sub(t, 2048)
        assert torch.equal(t.unsqueeze(0), P.sincos_pos_embed(dim, grid)), key
    # random_masking on injected noise (ties included: argsort order of equal keys is whatever torch.argsort gives;
    # the fixture avoids ties except in a dedicated all-distinct-by-construction case)
    m = ns.prithvi.MaskedAutoencoderViT(**SMALL)
    for tag, (N, L, D, r) in {"a": (2, 16, 4, 0.75), "b": (3, 196, 2, 0.75), "c": (2, 196, 2, 0.0), "d": (2, 16, 4, 0.5)}.items():
        x = detgen.normal(f"rm.{tag}.x", (N, L, D), seed=5)
        noise = detgen.uniform(f"rm.{tag}.n", (N, L), 0.0, 1.0, seed=5)
        with ref_harness.injected_rand([noise]):
            xm, mask, ids = m.random_masking(x, r)
        out[f"rm:{tag}:shape"] = np.array([N, L, D])
        out[f"rm:{tag}:ratio"] = np.array([r])
        out[f"rm:{tag}:xm"] = xm.numpy().copy()
        out[f"rm:{tag}:mask"] = mask.to(torch.uint8).numpy()
        out[f"rm:{tag}:ids"] = ids.to(torch.int32).numpy()
    # patchify / unpatchify
    for tag, args in {"s1": SMALL, "s3": SMALL_T3}.items():
        mm = ns.prithvi.MaskedAutoencoderViT(**args)
        c = P.MaeCfg(**args)
        x = detgen.normal(f"pf.{tag}", (2, c.in_chans, c.num_frames, c.img_size, c.img_size), seed=6)
        pt = mm.patchify(x)
        assert torch.equal(mm.unpatchify(pt), x)
        out[f"pf:{tag}:ck"] = checks(pt)
        out[f"pf:{tag}:sub"] = sub(pt, 1024)
    np.savez_compressed(HERE / "prithvi_misc.npz", **out)
    print("prithvi_misc written")


def gen_prithvi(ns, only_seg=False):
    if not only_seg:
        gen_mae(ns)
    seg_case(ns, "small_eval", SEG_SMALL, 2, 4, 8, True, False, 21, False)
    seg_case(ns, "small_train_frozen", SEG_SMALL, 2, 4, 8, True, True, 22, True)
    seg_case(ns, "small_train_unfrozen", SEG_SMALL, 2, 4, 8, False, True, 23, True)
    seg_case(ns, "full_eval_bs1", FULL, 1, 4, 256, True, False, 24, False)


def gen_mae(ns):
    gen_misc(ns)
    mae_case(ns, "small_bs2", SMALL, 2, 0.75, 11, True)
    mae_case(ns, "small_t3_bs2", SMALL_T3, 2, 0.75, 12, True)
    mae_case(ns, "small_r0_bs2", SMALL, 2, 0.0, 13, False)
    mae_case(ns, "full_bs1", FULL, 1, 0.75, 14, False)

"""CPU: the Prithvi oracle (oracle/prithvi_ref.py) reproduces the imported reference's outputs.

The transformer block body is timm's (third party, absent): both the fixtures and the oracle use the same
restatement of it, so these tests pin everything AROUND the block (patch embed, position tables, masking,
decoder plumbing, loss, neck, head) and are "parity unpinned" for the block body itself; its arithmetic is
cross-checked against torch's own MultiheadAttention / SDPA in test_vit_block_independent_check.
"""
import numpy as np
import pytest
import torch

from oracle import detgen, losses_ref, vit_block_ref
from oracle import prithvi_ref as P
from tests.helpers import MAE_CASES, SEG_CASES, checks, load, mae_inputs, rel_err, seg_inputs, sub


def test_pos_embed_tables_exact():
    g = load("prithvi_misc.npz")
    for dim, grid in ((768, (1, 14, 14)), (512, (1, 14, 14)), (512, (3, 14, 14)), (32, (3, 2, 2)), (16, (1, 4, 4))):
        t = P.sincos_pos_embed(dim, grid)[0]
        key = f"pos:{dim}:{grid[0]}x{grid[1]}x{grid[2]}"
        assert np.array_equal(sub(t, 2048), g[key + ":sub"])
        assert np.allclose(checks(t), g[key + ":ck"], rtol=1e-12)


def test_random_masking_exact():
    g = load("prithvi_misc.npz")
    for tag in "abcd":
        N, L, D = g[f"rm:{tag}:shape"]
        r = float(g[f"rm:{tag}:ratio"][0])
        x = detgen.normal(f"rm.{tag}.x", (N, L, D), seed=5)
        noise = detgen.uniform(f"rm.{tag}.n", (N, L), 0.0, 1.0, seed=5)
        xm, mask, ids = P.random_masking(x, r, noise)
        assert np.array_equal(xm.numpy(), g[f"rm:{tag}:xm"])
        assert np.array_equal(mask.to(torch.uint8).numpy(), g[f"rm:{tag}:mask"])
        assert np.array_equal(ids.to(torch.int32).numpy(), g[f"rm:{tag}:ids"])


def test_patchify_roundtrip_and_golden():
    from tests.helpers import PRITHVI_SMALL, PRITHVI_SMALL_T3
    g = load("prithvi_misc.npz")
    for tag, args in {"s1": PRITHVI_SMALL, "s3": PRITHVI_SMALL_T3}.items():
        c = P.MaeCfg(**args)
        x = detgen.normal(f"pf.{tag}", (2, c.in_chans, c.num_frames, c.img_size, c.img_size), seed=6)
        pt = P.patchify(c, x)
        assert torch.equal(P.unpatchify(c, pt), x)
        assert np.array_equal(sub(pt, 1024), g[f"pf:{tag}:sub"])


@pytest.mark.parametrize("tag", list(MAE_CASES))
def test_mae_matches_reference(tag):
    g = load(f"prithvi_mae_{tag}.npz")
    cfg, sd, x, noise, ratio = mae_inputs(tag)
    grads = MAE_CASES[tag][4]
    if grads:
        for k, v in sd.items():
            if not k.endswith("pos_embed"):
                v.requires_grad_(True)
    with torch.set_grad_enabled(grads):
        loss, pred, mask = P.mae_forward(sd, cfg, x, ratio, noise)
        latent, _, ids = P.forward_encoder(sd, cfg, x, ratio, noise)
    assert np.array_equal(mask.to(torch.uint8).numpy(), g["mask"])
    assert np.array_equal(ids.to(torch.int32).numpy(), g["ids_restore"])
    assert rel_err(sub(pred, 4096), g["pred_sub"]) < 1e-5
    assert rel_err(sub(latent, 4096), g["latent_sub"]) < 1e-5
    if np.isnan(g["loss"][0]):
        assert torch.isnan(loss)      # mask_ratio 0: 0/0 in forward_loss, as in the reference (prithvi.py:349)
    else:
        assert abs(loss.item() - g["loss"][0]) < 1e-5 * abs(g["loss"][0])
    if grads:
        loss.backward()
        tot = sum(v.grad.double().pow(2).sum().item() for v in sd.values() if v.grad is not None)
        assert abs(tot - g["grad_total_sq"][0]) < 1e-4 * g["grad_total_sq"][0]
        for key in g.files:
            if key.startswith("grad:"):
                assert rel_err(sub(sd[key[5:]].grad, 512), g[key]) < 1e-4, key


@pytest.mark.parametrize("tag", list(SEG_CASES))
def test_seg_matches_reference(tag):
    g = load(f"prithvi_seg_{tag}.npz")
    cfg, sd, x, y, noise, drop_u, train = seg_inputs(tag)
    want_grads = train
    if want_grads:
        for k, v in sd.items():
            trainable = v.dtype.is_floating_point and not k.endswith(("pos_embed", "running_mean", "running_var"))
            if trainable and not (cfg.frozen_backbone and k.startswith("backbone.")):
                v.requires_grad_(True)
    newbuf = {}
    with torch.set_grad_enabled(want_grads):
        logits = P.seg_forward(sd, cfg, x, noise, training=train, drop_u=drop_u, new_buffers=newbuf)
    assert rel_err(sub(logits, 4096), g["logits_sub"]) < 2e-5
    if g["margin_min"][0] > 1e-4:
        assert np.array_equal(losses_ref.class_mask(logits).to(torch.uint8).numpy(), g["mask"])
    ce = losses_ref.cross_entropy(logits, y, ignore_index=0)
    assert abs(ce.item() - g["loss_ce"][0]) < 1e-5 * abs(g["loss_ce"][0])
    if train:
        ce.backward()
        tot = sum(v.grad.double().pow(2).sum().item() for v in sd.values() if v.grad is not None)
        assert abs(tot - g["grad_total_sq"][0]) < 1e-3 * g["grad_total_sq"][0]
        for key in g.files:
            if key == "grad:head.net.0.bias":   # conv bias in front of train-mode BN: analytically 0, rounding noise only
                assert np.abs(g[key]).max() < 1e-6 and sd[key[5:]].grad.abs().max() < 1e-6
            elif key.startswith("grad:"):
                assert rel_err(sub(sd[key[5:]].grad, 512), g[key]) < 2e-3, key
        assert rel_err(newbuf["head.net.1.running_mean"].numpy(), g["rm:head.net.1"]) < 1e-5
        assert rel_err(newbuf["head.net.1.running_var"].numpy(), g["rv:head.net.1"]) < 1e-5
        assert int(newbuf["head.net.1.num_batches_tracked"]) == int(g["nbt:head.net.1"][0])


def test_vit_block_independent_check():
    """The restated timm Block against torch's own packed-in-proj MultiheadAttention + LayerNorm + GELU."""
    torch.manual_seed(0)
    dim, heads, B, N = 32, 4, 2, 9
    blk = vit_block_ref.Block(dim, heads, 4.0, qkv_bias=True)
    mha = torch.nn.MultiheadAttention(dim, heads, bias=True, batch_first=True)
    with torch.no_grad():
        mha.in_proj_weight.copy_(blk.attn.qkv.weight)
        mha.in_proj_bias.copy_(blk.attn.qkv.bias)
        mha.out_proj.weight.copy_(blk.attn.proj.weight)
        mha.out_proj.bias.copy_(blk.attn.proj.bias)
    x = torch.randn(B, N, dim)
    with torch.no_grad():
        h = blk.norm1(x)
        ref = x + mha(h, h, h, need_weights=False)[0]
        ref = ref + blk.mlp.fc2(torch.nn.functional.gelu(blk.mlp.fc1(blk.norm2(ref))))
        got = blk(x)
        sd = {"b." + k: v for k, v in blk.state_dict().items()}
        fn = P.vit_block(sd, "b", x, heads)
    assert torch.allclose(got, ref, atol=1e-5)
    assert torch.allclose(fn, ref, atol=1e-5)

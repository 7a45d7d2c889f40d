"""TEST INFRASTRUCTURE ONLY — CPU oracle for the Prithvi MAE-ViT path (torch CPU, functional).

Restates, over a reference-named state dict (no nn.Module state), what the reference computes in
  src/modules/prithvi.py              PatchEmbed :84-127, pos-embed tables :22-81, random_masking
                                      :258-283, forward_encoder :285-305, forward_decoder :307-331,
                                      forward_loss :333-350, patchify/unpatchify :236-256
  src/modules/prithvi_segmentation.py Norm2d :11-20, neck :23-72, FCNHead :75-111, net :132-162
Pinned by tests/golden/prithvi_*.npz (generated from the imported reference by
tests/golden/make_golden.py) — EXCEPT the transformer block itself, which the reference takes from
timm (not vendored, not installed): that part follows oracle/vit_block_ref.py and is PARITY UNPINNED.

Randomness is injected: `noise` [B, L] replaces torch.rand in random_masking (:267), `drop_u`
[B, C] the Bernoulli draw of Dropout2d (one uniform per (sample, channel); channel kept iff u >= p).
Never imported by the product (only tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch
import torch.nn.functional as F


@dataclass
class MaeCfg:
    img_size: int = 224
    patch_size: int = 16
    num_frames: int = 1
    tubelet_size: int = 1
    in_chans: int = 6
    embed_dim: int = 768
    depth: int = 12
    num_heads: int = 12
    decoder_embed_dim: int = 512
    decoder_depth: int = 8
    decoder_num_heads: int = 16
    mlp_ratio: float = 4.0
    norm_pix_loss: bool = False

    @property
    def grid(self):
        g = self.img_size // self.patch_size
        return (self.num_frames // self.tubelet_size, g, g)

    @property
    def num_patches(self):
        t, h, w = self.grid
        return t * h * w

    @property
    def patch_dim(self):
        return self.tubelet_size * self.patch_size * self.patch_size * self.in_chans


PRITHVI_100M = dict(img_size=224, patch_size=16, tubelet_size=1, in_chans=6, embed_dim=768, depth=12, num_heads=12,
                    decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16)  # prithvi_config.yaml:2-13


# ---- fixed sin-cos position table (prithvi.py:22-81) --------------------------------------------
def _sincos_1d(dim: int, n: int) -> np.ndarray:
    # frequencies are formed in float32 exactly as the reference does (np.arange(dtype=float32) / (dim/2)),
    # the outer product with the int64 positions promotes to float64
    omega = np.arange(dim // 2, dtype=np.float32)
    omega /= dim / 2.0
    omega = 1.0 / 10000**omega
    ang = np.arange(n).reshape(-1)[:, None] * omega[None, :]
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def sincos_pos_embed(dim: int, grid, cls_token: bool = True) -> torch.Tensor:
    """[1, (1+)T*H*W, dim] float32; feature split w: 6/16, h: 6/16, t: 4/16 of dim (:64-66)."""
    assert dim % 16 == 0
    t, h, w = grid
    dw = dh = dim // 16 * 6
    dt = dim // 16 * 4
    ew = np.tile(_sincos_1d(dw, w), (t * h, 1))
    eh = np.tile(np.repeat(_sincos_1d(dh, h), w, axis=0), (t, 1))
    et = np.repeat(_sincos_1d(dt, t), h * w, axis=0)
    table = np.concatenate([ew, eh, et], axis=1)
    if cls_token:
        table = np.concatenate([np.zeros((1, dim)), table], axis=0)
    return torch.from_numpy(table).float().unsqueeze(0)


# ---- state-dict shapes ---------------------------------------------------------------------------
def _block_shapes(prefix: str, dim: int, hidden: int) -> dict:
    return {
        f"{prefix}.norm1.weight": (dim,), f"{prefix}.norm1.bias": (dim,),
        f"{prefix}.attn.qkv.weight": (3 * dim, dim), f"{prefix}.attn.qkv.bias": (3 * dim,),
        f"{prefix}.attn.proj.weight": (dim, dim), f"{prefix}.attn.proj.bias": (dim,),
        f"{prefix}.norm2.weight": (dim,), f"{prefix}.norm2.bias": (dim,),
        f"{prefix}.mlp.fc1.weight": (hidden, dim), f"{prefix}.mlp.fc1.bias": (hidden,),
        f"{prefix}.mlp.fc2.weight": (dim, hidden), f"{prefix}.mlp.fc2.bias": (dim,),
    }


def mae_state_shapes(c: MaeCfg, decoder: bool = True) -> dict:
    """Reference registration order (prithvi.py:152-190)."""
    L, D, Dd = c.num_patches, c.embed_dim, c.decoder_embed_dim
    s: dict = {"cls_token": (1, 1, D), "pos_embed": (1, L + 1, D)}
    if decoder:
        s["mask_token"] = (1, 1, Dd)
    s["decoder_pos_embed"] = (1, L + 1, Dd)
    s["patch_embed.proj.weight"] = (D, c.in_chans, c.tubelet_size, c.patch_size, c.patch_size)
    s["patch_embed.proj.bias"] = (D,)
    for i in range(c.depth):
        s.update(_block_shapes(f"blocks.{i}", D, int(D * c.mlp_ratio)))
    s["norm.weight"] = (D,)
    s["norm.bias"] = (D,)
    if decoder:
        s["decoder_embed.weight"] = (Dd, D)
        s["decoder_embed.bias"] = (Dd,)
        for i in range(c.decoder_depth):
            s.update(_block_shapes(f"decoder_blocks.{i}", Dd, int(Dd * c.mlp_ratio)))
        s["decoder_norm.weight"] = (Dd,)
        s["decoder_norm.bias"] = (Dd,)
        s["decoder_pred.weight"] = (c.patch_dim, Dd)
        s["decoder_pred.bias"] = (c.patch_dim,)
    return s


@dataclass
class SegCfg:
    mae: MaeCfg
    num_classes: int = 4
    fcn_out_channels: int = 256
    fcn_num_convs: int = 1
    fcn_dropout: float = 0.1
    frozen_backbone: bool = True

    @property
    def embed(self):
        return self.mae.embed_dim * self.mae.num_frames


def seg_state_shapes(c: SegCfg) -> dict:
    """backbone (decoder attributes removed, load_prithvi no_decoder=True, utils.py:62-96), neck, head."""
    s = {"backbone." + k: v for k, v in mae_state_shapes(c.mae, decoder=False).items()}
    E = c.embed
    for i in (0, 3, 4, 7):
        s[f"neck.feature_pyramid_net.{i}.weight"] = (E, E, 2, 2)
        s[f"neck.feature_pyramid_net.{i}.bias"] = (E,)
        if i in (0, 4):
            s[f"neck.feature_pyramid_net.{i + 1}.ln.weight"] = (E,)
            s[f"neck.feature_pyramid_net.{i + 1}.ln.bias"] = (E,)
    # registration order inside the Sequential is by index; rebuild sorted
    order = []
    for i in range(8):
        for suffix in ("weight", "bias", "ln.weight", "ln.bias"):
            k = f"neck.feature_pyramid_net.{i}.{suffix}"
            if k in s:
                order.append(k)
    out = {k: v for k, v in s.items() if not k.startswith("neck.")}
    for k in order:
        out[k] = s[k]
    cin = E
    idx = 0
    for _ in range(c.fcn_num_convs):
        out[f"head.net.{idx}.weight"] = (c.fcn_out_channels, cin, 3, 3)
        out[f"head.net.{idx}.bias"] = (c.fcn_out_channels,)
        for nm, shp in (("weight", (c.fcn_out_channels,)), ("bias", (c.fcn_out_channels,)),
                        ("running_mean", (c.fcn_out_channels,)), ("running_var", (c.fcn_out_channels,)),
                        ("num_batches_tracked", ())):
            out[f"head.net.{idx + 1}.{nm}"] = shp
        cin = c.fcn_out_channels
        idx += 3
    idx += 1  # Dropout2d
    out[f"head.net.{idx}.weight"] = (c.num_classes, cin, 1, 1)
    out[f"head.net.{idx}.bias"] = (c.num_classes,)
    return out


# ---- pieces ---------------------------------------------------------------------------------------
def patchify(c: MaeCfg, imgs: torch.Tensor) -> torch.Tensor:
    """[B,C,T,H,W] -> [B, L, tub*p*p*C], feature order (tub, p, q, c) with c fastest (:236-245)."""
    B, C, T, H, W = imgs.shape
    p, tub = c.patch_size, c.tubelet_size
    x = imgs.reshape(B, C, T // tub, tub, H // p, p, W // p, p)
    x = x.permute(0, 2, 4, 6, 3, 5, 7, 1)  # b t h w tub p q c
    return x.reshape(B, (T // tub) * (H // p) * (W // p), tub * p * p * C)


def unpatchify(c: MaeCfg, x: torch.Tensor) -> torch.Tensor:
    """inverse of patchify for T = tub... the reference infers h = w = img/p and t from L (:247-256)."""
    B, L, Dm = x.shape
    p, tub = c.patch_size, c.tubelet_size
    n = c.img_size // p
    t = L // (n * n)
    C = Dm // (tub * p * p)
    x = x.reshape(B, t, n, n, tub, p, p, C).permute(0, 7, 1, 4, 2, 5, 3, 6)  # b c t tub h p w q
    return x.reshape(B, C, t * tub, n * p, n * p)


def patch_embed(sd, c: MaeCfg, x: torch.Tensor, prefix: str = "") -> torch.Tensor:
    """Conv3d(kernel = stride = (tub, p, p)) -> flatten -> [B, L, D] (:118-127)."""
    y = F.conv3d(x, sd[prefix + "patch_embed.proj.weight"], sd[prefix + "patch_embed.proj.bias"],
                 stride=(c.tubelet_size, c.patch_size, c.patch_size))
    return y.flatten(2).transpose(1, 2)


def random_masking(x: torch.Tensor, mask_ratio: float, noise: torch.Tensor):
    """argsort(noise) shuffling; keep the int(L*(1-r)) smallest (:258-283).  r = 0 is a pure shuffle."""
    N, L, D = x.shape
    keep = int(L * (1 - mask_ratio))
    ids_shuffle = torch.argsort(noise, dim=1)
    ids_restore = torch.argsort(ids_shuffle, dim=1)
    ids_keep = ids_shuffle[:, :keep]
    xm = torch.gather(x, 1, ids_keep.unsqueeze(-1).expand(-1, -1, D))
    mask = torch.ones(N, L, dtype=x.dtype)
    mask[:, :keep] = 0
    mask = torch.gather(mask, 1, ids_restore)
    return xm, mask, ids_restore


def vit_block(sd, prefix: str, x: torch.Tensor, heads: int) -> torch.Tensor:
    """timm Block restated (parity unpinned, see oracle/vit_block_ref.py): pre-norm attention + MLP."""
    B, N, D = x.shape
    hd = D // heads
    h = F.layer_norm(x, (D,), sd[prefix + ".norm1.weight"], sd[prefix + ".norm1.bias"], 1e-5)
    qkv = F.linear(h, sd[prefix + ".attn.qkv.weight"], sd[prefix + ".attn.qkv.bias"])
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = torch.softmax((q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(hd)), dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, D)
    x = x + F.linear(o, sd[prefix + ".attn.proj.weight"], sd[prefix + ".attn.proj.bias"])
    h = F.layer_norm(x, (D,), sd[prefix + ".norm2.weight"], sd[prefix + ".norm2.bias"], 1e-5)
    h = F.gelu(F.linear(h, sd[prefix + ".mlp.fc1.weight"], sd[prefix + ".mlp.fc1.bias"]))
    return x + F.linear(h, sd[prefix + ".mlp.fc2.weight"], sd[prefix + ".mlp.fc2.bias"])


def forward_encoder(sd, c: MaeCfg, x: torch.Tensor, mask_ratio: float, noise: torch.Tensor, prefix: str = ""):
    t = patch_embed(sd, c, x, prefix)
    pos = sd[prefix + "pos_embed"]
    t = t + pos[:, 1:, :]
    t, mask, ids_restore = random_masking(t, mask_ratio, noise)
    cls = (sd[prefix + "cls_token"] + pos[:, :1, :]).expand(t.shape[0], -1, -1)
    t = torch.cat([cls, t], dim=1)
    for i in range(c.depth):
        t = vit_block(sd, f"{prefix}blocks.{i}", t, c.num_heads)
    D = c.embed_dim
    t = F.layer_norm(t, (D,), sd[prefix + "norm.weight"], sd[prefix + "norm.bias"], 1e-5)
    return t, mask, ids_restore


def forward_decoder(sd, c: MaeCfg, latent: torch.Tensor, ids_restore: torch.Tensor) -> torch.Tensor:
    x = F.linear(latent, sd["decoder_embed.weight"], sd["decoder_embed.bias"])
    B, n_vis, Dd = x.shape
    L = ids_restore.shape[1]
    fill = sd["mask_token"].expand(B, L + 1 - n_vis, -1)
    body = torch.cat([x[:, 1:, :], fill], dim=1)
    body = torch.gather(body, 1, ids_restore.unsqueeze(-1).expand(-1, -1, Dd))
    x = torch.cat([x[:, :1, :], body], dim=1) + sd["decoder_pos_embed"]
    for i in range(c.decoder_depth):
        x = vit_block(sd, f"decoder_blocks.{i}", x, c.decoder_num_heads)
    x = F.layer_norm(x, (Dd,), sd["decoder_norm.weight"], sd["decoder_norm.bias"], 1e-5)
    x = F.linear(x, sd["decoder_pred.weight"], sd["decoder_pred.bias"])
    return x[:, 1:, :]


def forward_loss(c: MaeCfg, imgs: torch.Tensor, pred: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    target = patchify(c, imgs)
    if c.norm_pix_loss:
        mean = target.mean(dim=-1, keepdim=True)
        var = target.var(dim=-1, keepdim=True)
        target = (target - mean) / (var + 1.0e-6) ** 0.5
    per_patch = ((pred - target) ** 2).mean(dim=-1)
    return (per_patch * mask).sum() / mask.sum()


def mae_forward(sd, c: MaeCfg, imgs: torch.Tensor, mask_ratio: float, noise: torch.Tensor):
    """-> (loss, pred [B,L,patch_dim], mask [B,L]) (:352-356)."""
    latent, mask, ids_restore = forward_encoder(sd, c, imgs, mask_ratio, noise)
    pred = forward_decoder(sd, c, latent, ids_restore)
    return forward_loss(c, imgs, pred, mask), pred, mask


# ---- segmentation net ----------------------------------------------------------------------------
def norm2d(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """LayerNorm over channels of an NCHW map, eps 1e-6 (prithvi_segmentation.py:11-20)."""
    return F.layer_norm(x.permute(0, 2, 3, 1), (x.shape[1],), w, b, 1e-6).permute(0, 3, 1, 2)


def seg_forward(sd, c: SegCfg, x: torch.Tensor, noise: torch.Tensor, training: bool = False,
                drop_u: torch.Tensor | None = None, new_buffers: dict | None = None) -> torch.Tensor:
    """PrithviSegmentationNet.forward (:156-162): encoder at mask_ratio 0 (a pure token SHUFFLE by `noise`,
    which the neck then lays out row-major as if it were ordered — reference behaviour, kept), neck, head.
    `training` only affects the head's BatchNorm / Dropout2d when the backbone is frozen (it is put in eval
    mode, which changes nothing: the ViT has no dropout / BN)."""
    m = c.mae
    feats, _, _ = forward_encoder(sd, m, x, 0.0, noise, prefix="backbone.")
    t = feats[:, 1:, :]
    g = m.img_size // m.patch_size
    B = t.shape[0]
    t = t.reshape(B, g, g, -1).permute(0, 3, 1, 2)
    nk = "neck.feature_pyramid_net."
    t = F.conv_transpose2d(t, sd[nk + "0.weight"], sd[nk + "0.bias"], stride=2)
    t = F.gelu(norm2d(t, sd[nk + "1.ln.weight"], sd[nk + "1.ln.bias"]))
    t = F.conv_transpose2d(t, sd[nk + "3.weight"], sd[nk + "3.bias"], stride=2)
    t = F.conv_transpose2d(t, sd[nk + "4.weight"], sd[nk + "4.bias"], stride=2)
    t = F.gelu(norm2d(t, sd[nk + "5.ln.weight"], sd[nk + "5.ln.bias"]))
    t = F.conv_transpose2d(t, sd[nk + "7.weight"], sd[nk + "7.bias"], stride=2)
    idx = 0
    for _ in range(c.fcn_num_convs):
        t = F.conv2d(t, sd[f"head.net.{idx}.weight"], sd[f"head.net.{idx}.bias"], padding=1)
        bn = f"head.net.{idx + 1}"
        if training:
            rm, rv = sd[bn + ".running_mean"].clone(), sd[bn + ".running_var"].clone()
            t = F.batch_norm(t, rm, rv, sd[bn + ".weight"], sd[bn + ".bias"], True, 0.1, 1e-5)
            if new_buffers is not None:
                new_buffers[bn + ".running_mean"] = rm
                new_buffers[bn + ".running_var"] = rv
                new_buffers[bn + ".num_batches_tracked"] = sd[bn + ".num_batches_tracked"] + 1
        else:
            t = F.batch_norm(t, sd[bn + ".running_mean"], sd[bn + ".running_var"], sd[bn + ".weight"], sd[bn + ".bias"],
                             False, 0.1, 1e-5)
        t = F.relu(t)
        idx += 3
    if training and c.fcn_dropout > 0:
        assert drop_u is not None, "Dropout2d draw must be injected"
        keep = (drop_u >= c.fcn_dropout).to(t.dtype) / (1.0 - c.fcn_dropout)
        t = t * keep[:, :, None, None]
    idx += 1
    return F.conv2d(t, sd[f"head.net.{idx}.weight"], sd[f"head.net.{idx}.bias"])

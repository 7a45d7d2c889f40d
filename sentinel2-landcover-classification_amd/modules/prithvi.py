"""Prithvi MAE-ViT with the reference's constructor surface, running on the s2k HIP engine.

Drop-in for /root/reference/src/modules/prithvi.py:
  get_3d_sincos_pos_embed (:64-81)      fixed position tables (float64 numpy -> float32, bit-identical)
  PatchEmbed              (:84-127)     parameter holder (`proj` = Conv3d weight [D, C, tub, p, p])
  MaskedAutoencoderViT    (:130-356)    forward(imgs[B,C,T,H,W], mask_ratio) -> (loss, pred[B,L,p*p*C], mask[B,L]),
                                        forward_encoder -> (latent, mask, ids_restore), patchify / unpatchify
`state_dict()` keys, shapes and order equal the reference's.  The torch layers only hold parameters; forward and
backward are two native stage programs (plan/vit_plan.py -> csrc/), there is no CPU / ATen fallback.  The transformer
block is timm's `Block` in the reference (third party, absent here): its arithmetic follows the published
algorithm (oracle/vit_block_ref.py; parity unpinned at that boundary).

Randomness: `random_masking` draws `torch.rand(N, L)` in the reference (:267); here the same uniform noise is drawn
on the device, or injected through `self.masking_noise` ([B, L]) for parity tests.
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn

from ..flat import FlatParamsMixin
from ..plan.vit_plan import MaeSpec, mae_layout, plan_mae


def _sincos_1d(dim: int, n: int) -> np.ndarray:
    omega = np.arange(dim // 2, dtype=np.float32)
    omega /= dim / 2.0
    omega = 1.0 / 10000**omega
    ang = np.arange(n).reshape(-1)[:, None] * omega[None, :]
    return np.concatenate([np.sin(ang), np.cos(ang)], axis=1)


def get_3d_sincos_pos_embed(embed_dim: int, grid_size, cls_token: bool = False) -> np.ndarray:
    """[(1+)T*H*W, D] float64; features split w 6/16, h 6/16, t 4/16 of D (reference :64-81)."""
    assert embed_dim % 16 == 0
    t, h, w = grid_size
    dw = dh = embed_dim // 16 * 6
    dt = embed_dim // 16 * 4
    ew = np.tile(_sincos_1d(dw, w), (t * h, 1))
    eh = np.tile(np.repeat(_sincos_1d(dh, h), w, axis=0), (t, 1))
    et = np.repeat(_sincos_1d(dt, t), h * w, axis=0)
    table = np.concatenate([ew, eh, et], axis=1)
    if cls_token:
        table = np.concatenate([np.zeros([1, embed_dim]), table], axis=0)
    return table


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the s2k engine runs the whole network, sub-modules are not callable")


class PatchEmbed(_Holder):
    def __init__(self, img_size=224, patch_size=16, num_frames=3, tubelet_size=1, in_chans=3, embed_dim=768, norm_layer=None,
                 flatten=True, bias=True):
        super().__init__()
        self.img_size = (img_size, img_size) if isinstance(img_size, int) else tuple(img_size)
        self.patch_size = (patch_size, patch_size) if isinstance(patch_size, int) else tuple(patch_size)
        self.num_frames, self.tubelet_size = num_frames, tubelet_size
        self.grid_size = (num_frames // tubelet_size, self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1] * self.grid_size[2]
        self.flatten = flatten
        k = (tubelet_size, self.patch_size[0], self.patch_size[1])
        self.proj = nn.Conv3d(in_chans, embed_dim, kernel_size=k, stride=k, bias=bias)
        self.norm = nn.Identity()


class _Attention(_Holder):
    def __init__(self, dim, num_heads):
        super().__init__()
        self.num_heads = num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=True)
        self.proj = nn.Linear(dim, dim)


class _Mlp(_Holder):
    def __init__(self, dim, hidden):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)


class Block(_Holder):
    """Holder with timm's Block parameter names (norm1, attn.qkv, attn.proj, norm2, mlp.fc1, mlp.fc2)."""

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=True, norm_layer=nn.LayerNorm):
        super().__init__()
        self.norm1 = nn.LayerNorm(dim)
        self.attn = _Attention(dim, num_heads)
        self.norm2 = nn.LayerNorm(dim)
        self.mlp = _Mlp(dim, int(dim * mlp_ratio))


class MaskedAutoencoderViT(FlatParamsMixin, nn.Module):
    def __init__(self, img_size=224, patch_size=16, num_frames=3, tubelet_size=1, in_chans=3, embed_dim=1024, depth=24, num_heads=16,
                 decoder_embed_dim=512, decoder_depth=8, decoder_num_heads=16, mlp_ratio=4.0, norm_layer=nn.LayerNorm,
                 norm_pix_loss=False, _decoder: bool = True, _flat: bool = True):
        super().__init__()
        if norm_layer is not nn.LayerNorm:
            raise ValueError("only nn.LayerNorm is on the HIP path (the reference never passes anything else)")
        self.spec = MaeSpec(img_size, patch_size, num_frames, tubelet_size, in_chans, embed_dim, depth, num_heads, decoder_embed_dim,
                            decoder_depth, decoder_num_heads, mlp_ratio, norm_pix_loss, _decoder)
        self.patch_embed = PatchEmbed(img_size, patch_size, num_frames, tubelet_size, in_chans, embed_dim)
        num_patches = self.patch_embed.num_patches
        self.cls_token = nn.Parameter(torch.zeros(1, 1, embed_dim))
        self.pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, embed_dim), requires_grad=False)
        self.blocks = nn.ModuleList([Block(embed_dim, num_heads, mlp_ratio) for _ in range(depth)])
        self.norm = nn.LayerNorm(embed_dim)
        if _decoder:
            self.decoder_embed = nn.Linear(embed_dim, decoder_embed_dim, bias=True)
            self.mask_token = nn.Parameter(torch.zeros(1, 1, decoder_embed_dim))
        self.decoder_pos_embed = nn.Parameter(torch.zeros(1, num_patches + 1, decoder_embed_dim), requires_grad=False)
        if _decoder:
            self.decoder_blocks = nn.ModuleList([Block(decoder_embed_dim, decoder_num_heads, mlp_ratio) for _ in range(decoder_depth)])
            self.decoder_norm = nn.LayerNorm(decoder_embed_dim)
            self.decoder_pred = nn.Linear(decoder_embed_dim, tubelet_size * patch_size * patch_size * in_chans, bias=True)
        self.norm_pix_loss = norm_pix_loss
        self.initialize_weights()
        self.masking_noise: torch.Tensor | None = None   # inject [B, L] uniforms for parity tests
        self._unused_params = {"pos_embed", "decoder_pos_embed"}
        self._plans: dict = {}
        if _flat:
            self._init_flat(mae_layout(self.spec))

    # -- initialisation (reference :195-234) ------------------------------------------------------------
    @torch.no_grad()
    def reinitialize_pos_embed(self):
        g = self.patch_embed.grid_size
        self.pos_embed.copy_(torch.from_numpy(get_3d_sincos_pos_embed(self.pos_embed.shape[-1], g, cls_token=True)).float().unsqueeze(0))
        if hasattr(self, "decoder_pos_embed"):
            self.decoder_pos_embed.copy_(
                torch.from_numpy(get_3d_sincos_pos_embed(self.decoder_pos_embed.shape[-1], g, cls_token=True)).float().unsqueeze(0))

    @torch.no_grad()
    def initialize_weights(self):
        self.reinitialize_pos_embed()
        w = self.patch_embed.proj.weight.data
        nn.init.xavier_uniform_(w.view([w.shape[0], -1]))
        nn.init.normal_(self.cls_token, std=0.02)
        if hasattr(self, "mask_token"):
            nn.init.normal_(self.mask_token, std=0.02)
        self.apply(self._init_weights)

    def _init_weights(self, m):
        if isinstance(m, nn.Linear):
            nn.init.xavier_uniform_(m.weight)
            if m.bias is not None:
                nn.init.constant_(m.bias, 0)
        elif isinstance(m, nn.LayerNorm):
            nn.init.constant_(m.bias, 0)
            nn.init.constant_(m.weight, 1.0)

    # -- layout helpers (pure indexing, reference :236-256) ---------------------------------------------
    def patchify(self, imgs: torch.Tensor) -> torch.Tensor:
        s = self.spec
        B, C, T, H, W = imgs.shape
        p, tub = s.patch_size, s.tubelet_size
        x = imgs.reshape(B, C, T // tub, tub, H // p, p, W // p, p).permute(0, 2, 4, 6, 3, 5, 7, 1)
        return x.reshape(B, (T // tub) * (H // p) * (W // p), tub * p * p * C)

    def unpatchify(self, x: torch.Tensor) -> torch.Tensor:
        s = self.spec
        B, L, Dm = x.shape
        p, tub = s.patch_size, s.tubelet_size
        n = s.img_size // p
        t = L // (n * n)
        C = Dm // (tub * p * p)
        return x.reshape(B, t, n, n, tub, p, p, C).permute(0, 7, 1, 4, 2, 5, 3, 6).reshape(B, C, t * tub, n * p, n * p)

    # -- engine -------------------------------------------------------------------------------------------
    def _make_plan(self, B: int, training: bool, mask_ratio: float = 0.75, want_bwd: bool | None = None, want_dx: bool = False):
        return plan_mae(self.spec, B, mask_ratio, training, self._layout,    # (no BatchNorm / dropout: training == want_bwd)
                        bucket_floats=getattr(self, "_bucket_floats", 8 << 20), want_dx=want_dx, bf16=self.precision == "bf16-mixed")

    def _check_imgs(self, imgs):
        s = self.spec
        want = (s.in_chans, s.num_frames, s.img_size, s.img_size)
        if imgs.dim() != 5 or tuple(imgs.shape[1:]) != want:
            raise ValueError(f"expected [B,{want[0]},{want[1]},{want[2]},{want[3]}], got {tuple(imgs.shape)}")

    def _check(self, imgs):
        self._check_imgs(imgs)
        if not self.spec.decoder:
            raise RuntimeError("this MaskedAutoencoderViT was loaded without its decoder (load_prithvi(no_decoder=True))")

    def forward(self, imgs: torch.Tensor, mask_ratio: float = 0.75):
        self._check(imgs)
        if torch.compiler.is_compiling():      # the reference compiles `self.net` (train_mae_prithvi.py:59-64): one opaque node
            from ..compile_ops import compiled_mae_forward

            return compiled_mae_forward(self, imgs, mask_ratio)
        from ..vit_engine import run_vit

        out = run_vit(self, imgs, dict(noise=self.masking_noise), mask_ratio=mask_ratio)
        return out["loss"].reshape(()), out["pred"], out["mask"]

    def forward_encoder(self, x: torch.Tensor, mask_ratio: float):
        """(latent [B, 1 + keep, D], mask [B, L], ids_restore [B, L])   (reference prithvi.py:285-305); differentiable: the
        gradient of `latent` flows into the encoder's parameters.  Also runs on a decoder-less backbone (load_prithvi with
        no_decoder=True — the configuration in which the reference calls it, prithvi_segmentation.py:159)."""
        from ..plan.vit_plan import plan_mae_encoder
        from ..vit_engine import run_method

        self._check_imgs(x)
        mr = float(mask_ratio)
        wdx = torch.is_grad_enabled() and x.requires_grad
        out = run_method(self, ("encoder", tuple(x.shape), mr, wdx),
                         lambda bwd: plan_mae_encoder(self.spec, x.shape[0], mr, bwd, self._layout, want_dx=wdx and bwd),
                         {"x": x.contiguous()}, {"noise": self.masking_noise})
        return out["latent"], out["mask"], out["ids_restore"]

    def forward_decoder(self, x: torch.Tensor, ids_restore: torch.Tensor):
        """pred [B, L, tubelet * p * p * C] from the encoder's latent x [B, 1 + keep, D] and ids_restore [B, L]
        (reference prithvi.py:307-331); differentiable w.r.t. x and the decoder's parameters."""
        from ..plan.vit_plan import plan_mae_decoder
        from ..vit_engine import run_method

        if not self.spec.decoder:
            raise RuntimeError("this MaskedAutoencoderViT was loaded without its decoder (load_prithvi(no_decoder=True))")
        s = self.spec
        if x.dim() != 3 or x.shape[2] != s.embed_dim or tuple(ids_restore.shape) != (x.shape[0], s.num_patches):
            raise ValueError(f"expected x [B, 1 + keep, {s.embed_dim}] and ids_restore [B, {s.num_patches}], got {tuple(x.shape)}, {tuple(ids_restore.shape)}")
        B, N = x.shape[0], x.shape[1]
        out = run_method(self, ("decoder", B, N), lambda bwd: plan_mae_decoder(s, B, N, bwd, self._layout),
                         {"x": x.contiguous(), "ids_restore": ids_restore.contiguous()})
        return out["pred"]

    def forward_loss(self, imgs: torch.Tensor, pred: torch.Tensor, mask: torch.Tensor):
        """Masked mean squared error per patch (reference prithvi.py:333-350); differentiable w.r.t. pred."""
        from ..plan.vit_plan import plan_mae_loss
        from ..vit_engine import run_method

        self._check_imgs(imgs)
        s = self.spec
        B = imgs.shape[0]
        if tuple(pred.shape) != (B, s.num_patches, s.patch_dim) or tuple(mask.shape) != (B, s.num_patches):
            raise ValueError(f"expected pred [B, {s.num_patches}, {s.patch_dim}] and mask [B, {s.num_patches}]")
        out = run_method(self, ("loss", B), lambda bwd: plan_mae_loss(s, B, bwd, self._layout),
                         {"imgs": imgs.contiguous(), "pred": pred.contiguous(), "mask": mask.contiguous()}, uses_params=False)
        return out["loss"].reshape(())

    def random_masking(self, x: torch.Tensor, mask_ratio: float):
        """Per-sample random masking by argsort of uniform noise (reference prithvi.py:258-283): (x_masked [N, keep, D],
        mask [N, L] with 1 = removed, ids_restore [N, L]).  The noise is `self.masking_noise` if set, else drawn on the device."""
        from ..plan.vit_plan import plan_random_masking
        from ..vit_engine import run_method

        if x.dim() != 3:
            raise ValueError(f"expected x [N, L, D], got {tuple(x.shape)}")
        N, L, Dm = x.shape
        mr = float(mask_ratio)
        out = run_method(self, ("masking", N, L, Dm, mr), lambda bwd: plan_random_masking(self.spec, N, L, Dm, mr, bwd, self._layout),
                         {"x": x.contiguous()}, {"noise": self.masking_noise}, uses_params=False)
        return out["x_masked"], out["mask"], out["ids_restore"]

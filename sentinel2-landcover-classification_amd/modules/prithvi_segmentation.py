"""PrithviSegmentationNet with the reference's constructor surface, running on the s2k HIP engine.

Drop-in for /root/reference/src/modules/prithvi_segmentation.py:
  Norm2d (:11-20), ConvTransformerTokensToEmbeddingNeck (:23-72), FCNHead (:75-111)   parameter holders
  PrithviSegmentationNetConfig (:114-129), PrithviSegmentationNet (:132-162)
forward(x[B,C,T,H,W]) -> logits[B,num_classes,16*gh,16*gw].  Reference behaviour that is kept on purpose: the
backbone runs `forward_encoder(x, mask_ratio=0.0)`, which still SHUFFLES the tokens (argsort of fresh noise,
prithvi.py:258-283) and the neck then lays them out row-major — so the spatial arrangement fed to the neck is a
random permutation per call.  Noise is drawn on the device or injected (`masking_noise`, `dropout_noise`).
"""
from __future__ import annotations

from dataclasses import dataclass

import torch
from torch import nn

from ..flat import FlatParamsMixin
from ..plan.vit_plan import SegSpec, plan_seg, seg_layout
from .prithvi import MaskedAutoencoderViT, _Holder


class Norm2d(_Holder):
    def __init__(self, embed_dim: int):
        super().__init__()
        self.ln = nn.LayerNorm(embed_dim, eps=1e-6)


class ConvTransformerTokensToEmbeddingNeck(_Holder):
    def __init__(self, embed_dim: int, output_embed_dim: int, patch_height: int = 14, patch_width: int = 14, drop_cls_token: bool = True):
        super().__init__()
        if embed_dim != output_embed_dim:
            raise ValueError("the reference only builds the neck with output_embed_dim == embed_dim * num_frames")
        self.drop_cls_token, self.patch_height, self.patch_width = drop_cls_token, patch_height, patch_width
        ct = lambda i, o: nn.ConvTranspose2d(i, o, kernel_size=2, stride=2)  # noqa: E731
        self.feature_pyramid_net = nn.Sequential(ct(embed_dim, output_embed_dim), Norm2d(output_embed_dim), nn.GELU(),
                                                 ct(output_embed_dim, output_embed_dim), ct(output_embed_dim, output_embed_dim),
                                                 Norm2d(output_embed_dim), nn.GELU(), ct(output_embed_dim, output_embed_dim))


class FCNHead(_Holder):
    def __init__(self, num_classes: int, in_channels: int, out_channels: int, num_convs: int, dropout: float, kernel_size: int = 3):
        super().__init__()
        if kernel_size != 3:
            raise ValueError("only the 3x3 head convolution is on the HIP path")
        layers = []
        for i in range(num_convs):
            layers += [nn.Conv2d(in_channels if i == 0 else out_channels, out_channels, kernel_size, padding=kernel_size // 2),
                       nn.BatchNorm2d(out_channels), nn.ReLU(inplace=True)]
        self.net = nn.Sequential(*layers, nn.Dropout2d(dropout), nn.Conv2d(out_channels, num_classes, kernel_size=1))


@dataclass
class PrithviSegmentationNetConfig:
    num_frames: int
    num_classes: int
    fcn_out_channels: int
    fcn_num_convs: int
    fcn_dropout: float
    frozen_backbone: bool
    embed_dim: int = 768
    output_embed_dim: int = -1
    patch_height: int = 14
    patch_width: int = 14

    def __post_init__(self) -> None:
        self.output_embed_dim = self.embed_dim * self.num_frames


def initialize_head_or_neck_weights(m: nn.Module) -> None:
    """Reference :165-176."""
    if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
        nn.init.kaiming_normal_(m.weight, mode="fan_out", nonlinearity="relu")
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.BatchNorm2d):
        nn.init.constant_(m.weight, 1)
        nn.init.constant_(m.bias, 0)
    elif isinstance(m, nn.Linear):
        nn.init.xavier_normal_(m.weight)
        if m.bias is not None:
            nn.init.constant_(m.bias, 0)


class PrithviSegmentationNet(FlatParamsMixin, nn.Module):
    def __init__(self, config: PrithviSegmentationNetConfig, backbone: MaskedAutoencoderViT | None = None) -> None:
        """`backbone`: an (unflattened, decoder-less) MaskedAutoencoderViT; default = `utils.load_prithvi(num_frames)`
        exactly as the reference does (:135) — which needs weights/Prithvi_100M.pt."""
        super().__init__()
        from ..utils import load_prithvi

        self.config = config
        self.backbone = backbone if backbone is not None else load_prithvi(num_frames=config.num_frames, _flat=False)
        ms = self.backbone.spec
        if ms.decoder or self.backbone._layout is not None:
            raise ValueError("the backbone must be built with no_decoder=True and left unflattened (_flat=False)")
        if ms.embed_dim != config.embed_dim or ms.num_frames != config.num_frames:
            raise ValueError("backbone / config mismatch (embed_dim, num_frames)")
        self.neck = ConvTransformerTokensToEmbeddingNeck(config.embed_dim * config.num_frames, config.output_embed_dim,
                                                         config.patch_height, config.patch_width)
        self.head = FCNHead(config.num_classes, config.output_embed_dim, config.fcn_out_channels, config.fcn_num_convs, config.fcn_dropout)
        self.head.apply(initialize_head_or_neck_weights)
        self.neck.apply(initialize_head_or_neck_weights)
        if config.frozen_backbone:
            self.backbone.requires_grad_(False)
            self.backbone.eval()
        self.spec = SegSpec(ms, config.num_classes, config.fcn_out_channels, config.fcn_num_convs, config.fcn_dropout, config.frozen_backbone)
        self.masking_noise: torch.Tensor | None = None    # [B, L] uniforms of the (mask_ratio 0) token shuffle
        self.dropout_noise: torch.Tensor | None = None    # [B, fcn_out_channels] uniforms of Dropout2d (kept iff u >= p)
        layout = seg_layout(self.spec)
        self._unused_params = {n for n in layout.params if n in layout.frozen or (config.frozen_backbone and n.startswith("backbone."))}
        self._init_flat(layout)

    def _make_plan(self, B: int, training: bool, mask_ratio: float = 0.0, want_bwd: bool | None = None, want_dx: bool = False):
        return plan_seg(self.spec, B, training, self._layout, want_bwd=want_bwd, bucket_floats=getattr(self, "_bucket_floats", 8 << 20),
                        want_dx=want_dx, bf16=self.precision == "bf16-mixed")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        from ..vit_engine import run_vit

        m = self.spec.mae
        want = (m.in_chans, m.num_frames, m.img_size, m.img_size)
        if x.dim() != 5 or tuple(x.shape[1:]) != want:
            raise ValueError(f"expected [B,{want[0]},{want[1]},{want[2]},{want[3]}] (B,C,T,H,W), got {tuple(x.shape)}")
        if torch.compiler.is_compiling():
            from ..compile_ops import compiled_seg_forward

            return compiled_seg_forward(self, x)
        return run_vit(self, x, dict(noise=self.masking_noise, drop_u=self.dropout_noise))["logits"]

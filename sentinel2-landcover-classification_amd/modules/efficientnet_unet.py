"""EfficientNet-UNet with the reference's constructor surface, running on the s2k HIP engine.

Drop-in for /root/reference/src/modules/efficientnet_unet.py:
  EfficientNetConfig  (:18-53)   same fields / version table / bn_momentum flip
  EfficientNet        (:179-263) parameter holder with the same sub-module names
  EfficientnetUnet    (:106-166) forward(x[B,C,H,W]) -> logits[B,num_classes,H,W]
`state_dict()` keys, shapes and registration order equal the reference's, so
checkpoints move both ways.  The torch layers below only *hold* parameters: forward and backward
run as two native stage programs (plan/unet_plan.py -> csrc/), one C-ABI call each; there is no
CPU or ATen fallback — without the HIP library or a GPU tensor, forward raises.

Generalisations over the reference, identical at its native 224x224x6:
skip maps are dropped when their size equals the conv_head output's (reference: literal (7, 7),
:259) and size[4] = 32 + in_channels (reference: 38, :154-165); H, W must be multiples of 32.
"""
from __future__ import annotations

import math
import typing
from dataclasses import dataclass

import torch
from torch import nn

from ..plan import opdefs as D
from ..plan.unet_plan import BlockSpec, UnetSpec, build_layout, plan_unet
from ..compile_ops import compiled_unet_forward
from ..flat import FlatParamsMixin

_VERSION_TABLE = {  # (width, depth, dropout)   efficientnet_unet.py:35-45
    "b0": (1.0, 1.0, 0.2), "b1": (1.0, 1.1, 0.2), "b2": (1.1, 1.2, 0.3), "b3": (1.2, 1.4, 0.3),
    "b4": (1.4, 1.8, 0.4), "b5": (1.6, 2.2, 0.4), "b6": (1.8, 2.6, 0.5), "b7": (2.0, 3.1, 0.5),
}
_STAGE_STRINGS = [  # efficientnet_unet.py:202-208
    "r1_k3_s11_e1_i32_o16_se0.25", "r2_k3_s22_e6_i16_o24_se0.25", "r2_k5_s22_e6_i24_o40_se0.25",
    "r3_k3_s22_e6_i40_o80_se0.25", "r3_k5_s11_e6_i80_o112_se0.25", "r4_k5_s22_e6_i112_o192_se0.25",
    "r1_k3_s11_e6_i192_o320_se0.25",
]
_HEAD_CHANNELS = {"b0": 1280, "b1": 1280, "b2": 1408, "b3": 1536, "b4": 1792, "b5": 2048, "b6": 2304, "b7": 2560}


@dataclass
class EfficientNetConfig:
    version: typing.Literal["b0", "b1", "b2", "b3", "b4", "b5", "b6", "b7"]
    in_channels: int
    num_classes: int
    bn_momentum: float = 0.99
    bn_epsilon: float = 1e-3
    depth_divisor: int | None = 8
    drop_connect_rate: float | None = 0.2
    min_depth: int | None = None
    class_distribution: list[float] | None = None
    dropout_rate: float | None = None
    width_coefficient: float | None = None
    depth_coefficient: float | None = None

    def __post_init__(self) -> None:
        if self.version not in _VERSION_TABLE:
            raise ValueError(f"There is no model version {self.version}")
        w, d, p = _VERSION_TABLE[self.version]
        self.width_coefficient = self.width_coefficient or w
        self.depth_coefficient = self.depth_coefficient or d
        self.dropout_rate = self.dropout_rate or p
        self.bn_momentum = 1 - self.bn_momentum  # torch momentum convention, reference :53


def _round_filters(filters: int, width: float | None, divisor: int | None, min_depth: int | None) -> int:
    assert min_depth is not None or divisor is not None, "min_depth or depth_divisor should be supplied"
    if width is None:
        return filters
    scaled = filters * width
    floor = min_depth or divisor
    rounded = max(floor, int(scaled + divisor / 2) // divisor * divisor)
    if rounded < 0.9 * scaled:  # never round down by more than 10 %
        rounded += divisor
    return int(rounded)


def _parse_stage(s: str) -> dict:
    out = {"noskip": "noskip" in s}
    for tok in s.split("_"):
        if tok[:2] == "se":
            out["se"] = float(tok[2:])
        elif tok and tok[0] in "rkseio" and tok[1:].replace(".", "").isdigit():
            out[tok[0]] = tok[1:]
    if "s" not in out or len(out["s"]) != 2:
        raise ValueError("Strides options should be a pair of integers.")
    return out


def block_specs(config: EfficientNetConfig) -> list[BlockSpec]:
    """Width/depth scaling of the 7 stages (reference :199-226), including its stride quirk: the
    first block of a stage keeps a *tuple* stride, `(1, 1) == 1` is False (:383), so only repeat
    blocks ever take the residual."""
    rf = lambda f: _round_filters(f, config.width_coefficient, config.depth_divisor, config.min_depth)  # noqa: E731
    specs: list[BlockSpec] = []
    for s in _STAGE_STRINGS:
        o = _parse_stage(s)
        cin, cout = rf(int(o["i"])), rf(int(o["o"]))
        reps = int(math.ceil(config.depth_coefficient * int(o["r"]))) if config.depth_coefficient is not None else int(o["r"])
        k, e, stride = int(o["k"]), int(o["e"]), int(o["s"][0])
        se_ratio = o.get("se")
        skip = not o["noskip"]
        for r in range(reps):
            first = r == 0
            bi = cin if first else cout
            st = stride if first else 1
            se = max(1, int(bi * se_ratio)) if se_ratio and 0 < se_ratio <= 1 else 0
            specs.append(BlockSpec(k, st, bi, cout, e, se, skip and (not first) and bi == cout))
    return specs


def unet_spec(config: EfficientNetConfig) -> UnetSpec:
    rf = lambda f: _round_filters(f, config.width_coefficient, config.depth_divisor, config.min_depth)  # noqa: E731
    return UnetSpec(config.version, config.in_channels, config.num_classes, rf(32), rf(1280), block_specs(config),
                    config.bn_momentum, config.bn_epsilon, config.drop_connect_rate)


# ---------------------------------------------------------------------------------------------
# parameter holders (same module tree / names as the reference; never called)
# ---------------------------------------------------------------------------------------------

class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the s2k engine runs the whole network, sub-modules are not callable")


class _Flatten(_Holder):
    pass


class MBConvBlock(_Holder):
    def __init__(self, b: BlockSpec, mom: float, eps: float) -> None:
        super().__init__()
        self.skip_connection, self.stride = True, b.stride
        self.input_filters, self.output_filters = b.cin, b.cout
        mods: list[nn.Module] = []
        if b.expand != 1:
            mods += [nn.Conv2d(b.cin, b.cexp, 1, bias=False), nn.BatchNorm2d(b.cexp, momentum=mom, eps=eps), nn.SiLU()]
        mods += [nn.Conv2d(b.cexp, b.cexp, b.kernel, stride=b.stride, groups=b.cexp, bias=False),
                 nn.BatchNorm2d(b.cexp, momentum=mom, eps=eps), nn.SiLU()]
        self.stem = nn.Sequential(*mods)
        self.has_squeeze_excitation = b.se > 0
        if b.se > 0:
            self.squeeze_excitation = nn.Sequential(nn.AdaptiveAvgPool2d(1), nn.Conv2d(b.cexp, b.se, 1), nn.SiLU(),
                                                    nn.Conv2d(b.se, b.cexp, 1))
        self.final_layer = nn.Sequential(nn.Conv2d(b.cexp, b.cout, 1, bias=False),
                                         nn.BatchNorm2d(b.cout, momentum=mom, eps=eps))


class EfficientNet(FlatParamsMixin, nn.Module):
    """EfficientNet b0-b7 (reference :179-263): `encode(x) -> (x, feature_maps)`, `forward(x) -> logits [B, num_classes]`.
    Standalone it owns its flat parameter buffer; as the `encoder` of an EfficientnetUnet it is a view of the owner's (the U-Net
    never uses `fc`, which therefore never receives a gradient there — as in the reference)."""

    def __init__(self, config: EfficientNetConfig, _owner=None) -> None:
        super().__init__()
        self.name = config.version
        self.drop_connect_rate = config.drop_connect_rate
        self.dropout_rate = config.dropout_rate
        spec = unet_spec(config)
        self.spec = spec
        mom, eps = config.bn_momentum, config.bn_epsilon
        self.stem = nn.Sequential(nn.Conv2d(config.in_channels, spec.stem_out, 3, stride=2, bias=False),
                                  nn.BatchNorm2d(spec.stem_out, momentum=mom, eps=eps), nn.SiLU())
        self.blocks = nn.ModuleList([MBConvBlock(b, mom, eps) for b in spec.blocks])
        self.conv_head = nn.Sequential(nn.Conv2d(spec.blocks[-1].cout, spec.head_out, 1, bias=False),
                                       nn.BatchNorm2d(spec.head_out, momentum=mom, eps=eps), nn.SiLU())
        self.fc = nn.Sequential(nn.AdaptiveAvgPool2d(1), _Flatten(), nn.Dropout(p=config.dropout_rate),
                                nn.Linear(spec.head_out, config.num_classes))
        self.drop_connect_noise: torch.Tensor | None = None   # standalone: inject [n_blocks, B] uniforms for parity tests
        self.dropout_noise: torch.Tensor | None = None        # [B, head channels] uniforms of fc's Dropout (kept iff u >= p)
        self._owner_ref = None
        self._standalone = _owner is None
        if self._standalone:
            from ..plan.unet_plan import build_encoder_layout

            self.apply(init_weights)
            self._unused_params = set()
            self._init_flat(build_encoder_layout(spec))

    def _apply(self, fn, recurse=True):
        if self._standalone:
            return super()._apply(fn, recurse)
        return nn.Module._apply(self, fn, recurse)      # the owner re-flattens

    def _owner_prefix(self):
        if self._standalone:
            return self, ""
        owner = self._owner_ref() if self._owner_ref is not None else None
        if owner is None:
            raise RuntimeError("this EfficientNet was built as the encoder of an EfficientnetUnet that no longer exists")
        return owner, "encoder."

    def _run(self, x: torch.Tensor, classifier: bool) -> dict:
        from ..plan.encoder_plan import plan_encoder
        from ..vit_engine import run_method

        owner, pre = self._owner_prefix()
        if x.dim() != 4 or x.shape[1] != self.spec.in_channels:
            raise ValueError(f"expected [B,{self.spec.in_channels},H,W], got {tuple(x.shape)}")
        B, _, H, W = x.shape
        training = self.training
        wdx = torch.is_grad_enabled() and x.requires_grad
        p_drop = float(self.dropout_rate or 0.0)
        spec, layout = self.spec, owner._layout
        dc = owner.drop_connect_noise if not self._standalone else self.drop_connect_noise
        out = run_method(owner, ("efficientnet", pre, tuple(x.shape), training, classifier, wdx),
                         lambda bwd: plan_encoder(spec, B, H, W, training, layout, pre, classifier, bwd, wdx and bwd, p_drop),
                         {"x": x.contiguous()}, {"drop_connect": dc if training else None, "dropout_u": self.dropout_noise},
                         publish_all=classifier)
        if training:       # num_batches_tracked of the encoder's BatchNorms
            idx = getattr(self, "_nbt_idx", None)
            if idx is None or idx.device != owner._flat_nbt.device:
                idx = torch.tensor([i for i, n in enumerate(layout.nbt) if n.startswith(pre)], dtype=torch.long, device=owner._flat_nbt.device)
                self._nbt_idx = idx
            owner._flat_nbt[idx] += 1
        return out

    def encode(self, x: torch.Tensor):
        """(x, feature_maps): the activated conv_head output, and [x, first block output at each new spatial size, deepest
        first] (reference :251-263; the (7, 7) literal generalised: maps of the conv_head output's size are dropped)."""
        out = self._run(x, False)
        return out["x"], [out["x"]] + [out[f"f{k}"] for k in range(len(out) - 1)]

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """logits [B, num_classes] = fc(encode(x)[0]) (reference :246-249)."""
        return self._run(x, True)["logits"]


def _double_conv(cin: int, cout: int) -> nn.Sequential:
    return nn.Sequential(nn.Conv2d(cin, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True),
                         nn.Conv2d(cout, cout, 3, 1, 1), nn.BatchNorm2d(cout), nn.ReLU(inplace=True))


def init_weights(m: nn.Module) -> None:
    """Reference :401-412."""
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


class EfficientnetUnet(FlatParamsMixin, nn.Module):
    def __init__(self, config: EfficientNetConfig, concat_input: bool = True) -> None:
        super().__init__()
        if not concat_input:
            raise ValueError("concat_input=False cannot work in the reference either (out_conv1x1 expects 32 channels)")
        from ..utils import initialize_classification_layer_bias

        self.config = config
        self.spec = unet_spec(config)
        self.encoder = EfficientNet(config, _owner=self)
        ups_in = [self.n_channels, 512, 256, 128]
        ups_out = [512, 256, 128, 64]
        self.up_convs = nn.ModuleList([nn.ConvTranspose2d(i, o, kernel_size=2, stride=2) for i, o in zip(ups_in, ups_out)])
        self.double_convs = nn.ModuleList([_double_conv(c, o) for c, o in zip(self.size[:4], ups_out)])
        self.concat_input = concat_input
        self.input_up_conv = nn.ConvTranspose2d(64, 32, kernel_size=2, stride=2)
        self.input_double_conv = _double_conv(self.size[4], 32)
        self.out_conv1x1 = nn.Conv2d(self.size[5], config.num_classes, kernel_size=1)
        self.apply(init_weights)
        initialize_classification_layer_bias(self.out_conv1x1, class_distribution=config.class_distribution)
        self.drop_connect_noise: torch.Tensor | None = None  # inject [n_blocks, B] uniforms for parity tests
        self._unused_params = {"encoder.fc.3.weight", "encoder.fc.3.bias"}  # never on the U-Net path (as in the reference)
        self._init_flat(build_layout(self.spec))
        import weakref

        self.encoder._owner_ref = weakref.ref(self)

    @property
    def n_channels(self) -> int:
        return _HEAD_CHANNELS[self.encoder.name]

    @property
    def size(self) -> list[int]:
        from ..plan.unet_plan import CAT_SIZES

        return CAT_SIZES[self.encoder.name] + [32 + self.config.in_channels, 32]

    # -- engine ---------------------------------------------------------------------------
    def _make_plan(self, B: int, H: int, W: int, training: bool, want_bwd: bool | None = None, want_dx: bool = False):
        return plan_unet(self.spec, B, H, W, training, self._layout, defer_wgrads=getattr(self, "_defer_wgrads", None),
                         want_bwd=want_bwd, bucket_floats=getattr(self, "_bucket_floats", 8 << 20), want_dx=want_dx,
                         bf16=self.precision == "bf16-mixed")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4 or x.shape[1] != self.config.in_channels:
            raise ValueError(f"expected [B,{self.config.in_channels},H,W], got {tuple(x.shape)}")
        if torch.compiler.is_compiling():
            # the reference compiles `self.net` unless --type debug (train_segmentation.py:70-75): Dynamo sees ONE opaque node
            return compiled_unet_forward(self, x)
        from ..engine import run_unet

        return run_unet(self, x)

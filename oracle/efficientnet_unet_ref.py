"""TEST INFRASTRUCTURE ONLY — CPU oracle: EfficientNet-UNet forward (autograd gives backward).

Functional restatement, in plain torch CPU ops over a reference-named ``state_dict``, of
  /root/reference/src/modules/efficientnet_unet.py
    EfficientNetConfig.__post_init__ :34-53     (version table, bn_momentum flip)
    BlockConfig.from_str / block table  :83-103, :199-226
    _round_filters                      :266-278
    Conv2dSamePadding.forward           :288-297  (TF "SAME", asymmetric pad)
    MBConvBlock.forward                 :377-387  (incl. the tuple-vs-int stride quirk)
    _drop_connect                       :390-398
    EfficientNet.encode                 :251-263
    EfficientnetUnet.forward / size     :125-166, _double_conv :168-176
plus the two generalisations SURVEY.md §8 a7-G needs for BASELINE shapes (identical to the
reference at 224x224x6): feature maps whose spatial size equals the conv_head output's are
dropped (reference: literal ``(7, 7)``), and ``size[4] = 32 + in_channels`` (reference: 38).

Pinned by tests/golden/unet_*.npz (generated from the imported reference by make_golden.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field

import torch
import torch.nn.functional as F

_VERSIONS = {  # efficientnet_unet.py:35-45  (width, depth, resolution(unused), dropout)
    "b0": (1.0, 1.0, 224, 0.2), "b1": (1.0, 1.1, 240, 0.2), "b2": (1.1, 1.2, 260, 0.3),
    "b3": (1.2, 1.4, 300, 0.3), "b4": (1.4, 1.8, 380, 0.4), "b5": (1.6, 2.2, 456, 0.4),
    "b6": (1.8, 2.6, 528, 0.5), "b7": (2.0, 3.1, 600, 0.5),
}
# efficientnet_unet.py:202-208: (repeats, kernel, stride, expand, in, out); se_ratio .25 everywhere
_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80),
           (3, 5, 1, 6, 80, 112), (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
_HEAD = {"b0": 1280, "b1": 1280, "b2": 1408, "b3": 1536, "b4": 1792, "b5": 2048, "b6": 2304, "b7": 2560}
_CAT = {"b0": [592, 296, 152, 80], "b1": [592, 296, 152, 80], "b2": [600, 304, 152, 80],
        "b3": [608, 304, 160, 88], "b4": [624, 312, 160, 88], "b5": [640, 320, 168, 88],
        "b6": [656, 328, 168, 96], "b7": [672, 336, 176, 96]}


def round_filters(filters: int, width: float, divisor: int = 8) -> int:
    """efficientnet_unet.py:266-278."""
    f = filters * width
    new = max(divisor, int(f + divisor / 2) // divisor * divisor)
    if new < 0.9 * f:
        new += divisor
    return int(new)


@dataclass
class RefBlock:
    kernel: int
    stride: int
    cin: int
    cout: int
    expand: int
    se: int
    first_of_stage: bool  # keeps stride as a tuple in the reference -> never takes the residual

    @property
    def cexp(self) -> int:
        return self.cin * self.expand

    @property
    def residual(self) -> bool:
        # efficientnet_unet.py:383 `self.stride == 1`: a tuple (1, 1) != 1, so only repeat blocks
        # (int stride 1, :224-226) can pass; they always have cin == cout.
        return (not self.first_of_stage) and self.stride == 1 and self.cin == self.cout


@dataclass
class RefNet:
    version: str
    in_channels: int
    num_classes: int
    bn_momentum: float = 0.01  # config value 0.99 flipped at :53
    bn_eps: float = 1e-3
    drop_connect_rate: float | None = 0.2
    blocks: list[RefBlock] = field(default_factory=list)
    stem_out: int = 0
    head_out: int = 0


def build(version: str, in_channels: int, num_classes: int, bn_momentum: float = 0.99,
          bn_eps: float = 1e-3, drop_connect_rate: float | None = 0.2) -> RefNet:
    w, d, _, _ = _VERSIONS[version]
    net = RefNet(version, in_channels, num_classes, 1 - bn_momentum, bn_eps, drop_connect_rate)
    net.stem_out = round_filters(32, w)
    for rep, k, s, e, i, o in _STAGES:
        i, o = round_filters(i, w), round_filters(o, w)
        rep = int(math.ceil(d * rep))
        se = max(1, int(i * 0.25))  # :348 uses the stage's *input* filters of that block config
        net.blocks.append(RefBlock(k, s, i, o, e, se, True))
        for _ in range(rep - 1):
            # :221-226 — after the first block the shared config is mutated: in=out, stride=1
            net.blocks.append(RefBlock(k, 1, o, o, e, max(1, int(o * 0.25)), False))
    net.head_out = round_filters(1280, w)
    assert net.head_out == _HEAD[version]
    return net


def same_pad(x: torch.Tensor, k: int, s: int) -> torch.Tensor:
    """Conv2dSamePadding.forward :288-297 — left/top = pad//2, right/bottom = pad - pad//2."""
    h, w = x.shape[-2:]
    ph = max((math.ceil(h / s) - 1) * s + (k - 1) + 1 - h, 0)
    pw = max((math.ceil(w / s) - 1) * s + (k - 1) + 1 - w, 0)
    if ph > 0 or pw > 0:
        x = F.pad(x, [pw // 2, pw - pw // 2, ph // 2, ph - ph // 2])
    return x


class _BN:
    """Train/eval BatchNorm2d over a state-dict; collects running-stat updates on the side."""

    def __init__(self, sd, training: bool, new_buffers: dict | None):
        self.sd, self.training, self.new = sd, training, new_buffers

    def __call__(self, x, prefix: str, momentum: float, eps: float):
        sd = self.sd
        if not self.training:
            return F.batch_norm(x, sd[prefix + ".running_mean"], sd[prefix + ".running_var"],
                                sd[prefix + ".weight"], sd[prefix + ".bias"], False, momentum, eps)
        rm = sd[prefix + ".running_mean"].detach().clone()
        rv = sd[prefix + ".running_var"].detach().clone()
        y = F.batch_norm(x, rm, rv, sd[prefix + ".weight"], sd[prefix + ".bias"], True, momentum, eps)
        if self.new is not None:
            self.new[prefix + ".running_mean"] = rm
            self.new[prefix + ".running_var"] = rv
            self.new[prefix + ".num_batches_tracked"] = sd[prefix + ".num_batches_tracked"] + 1
        return y


def drop_connect(x, rate: float, noise_b: torch.Tensor):
    """_drop_connect :390-398 with the U[0,1) draw supplied by the caller (noise_b: [B])."""
    keep = 1.0 - rate
    binary = torch.floor(keep + noise_b.to(x.dtype)).view(-1, 1, 1, 1)
    return x / keep * binary


def mbconv(sd, bn: _BN, net: RefNet, idx: int, x, dc_rate, dc_noise_b):
    """MBConvBlock.forward :377-387."""
    b = net.blocks[idx]
    p = f"encoder.blocks.{idx}."
    identity = x
    j = 0
    if b.expand != 1:
        x = F.conv2d(x, sd[p + "stem.0.weight"])
        x = F.silu(bn(x, p + "stem.1", net.bn_momentum, net.bn_eps))
        j = 3
    x = F.conv2d(same_pad(x, b.kernel, b.stride), sd[p + f"stem.{j}.weight"], None, b.stride, 0, 1, b.cexp)
    x = F.silu(bn(x, p + f"stem.{j + 1}", net.bn_momentum, net.bn_eps))
    s = F.adaptive_avg_pool2d(x, 1)
    s = F.silu(F.conv2d(s, sd[p + "squeeze_excitation.1.weight"], sd[p + "squeeze_excitation.1.bias"]))
    s = F.conv2d(s, sd[p + "squeeze_excitation.3.weight"], sd[p + "squeeze_excitation.3.bias"])
    x = x * torch.sigmoid(s)
    x = F.conv2d(x, sd[p + "final_layer.0.weight"])
    x = bn(x, p + "final_layer.1", net.bn_momentum, net.bn_eps)
    if b.residual:
        if dc_rate and bn.training:  # :384 truthiness: rate 0.0 (block 0) and None skip it
            x = drop_connect(x, dc_rate, dc_noise_b)
        x = x + identity
    return x


def encode(sd, bn: _BN, net: RefNet, x, dc_noise):
    """EfficientNet.encode :251-263 (+ a7-G (i)).  dc_noise: [n_blocks, B] uniform draws or None."""
    x = F.conv2d(same_pad(x, 3, 2), sd["encoder.stem.0.weight"], None, 2)
    x = F.silu(bn(x, "encoder.stem.1", net.bn_momentum, net.bn_eps))
    candidates = []
    n = len(net.blocks)
    for i in range(n):
        rate = net.drop_connect_rate * (i / n) if net.drop_connect_rate is not None else None
        noise = dc_noise[i] if (dc_noise is not None) else None
        if rate and bn.training and net.blocks[i].residual and noise is None:
            raise ValueError("training-mode oracle needs injected drop-connect noise")
        x = mbconv(sd, bn, net, i, x, rate, noise)
        candidates.append(x)
    head_hw = x.shape[-2:]
    fmaps: list[torch.Tensor] = []
    for c in candidates:  # first block output at each new spatial size, deepest first
        if c.shape[-2:] not in [f.shape[-2:] for f in fmaps] and c.shape[-2:] != head_hw:
            fmaps.insert(0, c)
    x = F.conv2d(x, sd["encoder.conv_head.0.weight"])
    x = F.silu(bn(x, "encoder.conv_head.1", net.bn_momentum, net.bn_eps))
    fmaps.insert(0, x)
    return x, fmaps


def _double_conv(sd, bn: _BN, p: str, x):
    """_double_conv :168-176 (decoder BNs use torch defaults momentum .1, eps 1e-5)."""
    x = F.conv2d(x, sd[p + ".0.weight"], sd[p + ".0.bias"], 1, 1)
    x = F.relu(bn(x, p + ".1", 0.1, 1e-5))
    x = F.conv2d(x, sd[p + ".3.weight"], sd[p + ".3.bias"], 1, 1)
    x = F.relu(bn(x, p + ".4", 0.1, 1e-5))
    return x


def unet_forward(sd: dict, net: RefNet, x: torch.Tensor, training: bool = False,
                 dc_noise: torch.Tensor | None = None, new_buffers: dict | None = None) -> torch.Tensor:
    """EfficientnetUnet.forward :125-138."""
    bn = _BN(sd, training, new_buffers)
    identity = x
    _, fmaps = encode(sd, bn, net, x, dc_noise)
    x = fmaps.pop(0)
    for i, fm in enumerate(fmaps):
        x = F.conv_transpose2d(x, sd[f"up_convs.{i}.weight"], sd[f"up_convs.{i}.bias"], 2)
        x = torch.cat([x, fm], dim=1)
        x = _double_conv(sd, bn, f"double_convs.{i}", x)
    x = F.conv_transpose2d(x, sd["input_up_conv.weight"], sd["input_up_conv.bias"], 2)
    x = torch.cat([x, identity], dim=1)
    x = _double_conv(sd, bn, "input_double_conv", x)
    return F.conv2d(x, sd["out_conv1x1.weight"], sd["out_conv1x1.bias"])


def state_shapes(net: RefNet) -> dict[str, tuple]:
    """Names/shapes of the reference ``state_dict()``, in registration order."""
    out: dict[str, tuple] = {}

    def bn(p, c):
        out[p + ".weight"] = (c,); out[p + ".bias"] = (c,)
        out[p + ".running_mean"] = (c,); out[p + ".running_var"] = (c,)
        out[p + ".num_batches_tracked"] = ()

    out["encoder.stem.0.weight"] = (net.stem_out, net.in_channels, 3, 3)
    bn("encoder.stem.1", net.stem_out)
    for i, b in enumerate(net.blocks):
        p = f"encoder.blocks.{i}."
        j = 0
        if b.expand != 1:
            out[p + "stem.0.weight"] = (b.cexp, b.cin, 1, 1)
            bn(p + "stem.1", b.cexp)
            j = 3
        out[p + f"stem.{j}.weight"] = (b.cexp, 1, b.kernel, b.kernel)
        bn(p + f"stem.{j + 1}", b.cexp)
        out[p + "squeeze_excitation.1.weight"] = (b.se, b.cexp, 1, 1)
        out[p + "squeeze_excitation.1.bias"] = (b.se,)
        out[p + "squeeze_excitation.3.weight"] = (b.cexp, b.se, 1, 1)
        out[p + "squeeze_excitation.3.bias"] = (b.cexp,)
        out[p + "final_layer.0.weight"] = (b.cout, b.cexp, 1, 1)
        bn(p + "final_layer.1", b.cout)
    out["encoder.conv_head.0.weight"] = (net.head_out, net.blocks[-1].cout, 1, 1)
    bn("encoder.conv_head.1", net.head_out)
    out["encoder.fc.3.weight"] = (net.num_classes, net.head_out)
    out["encoder.fc.3.bias"] = (net.num_classes,)
    ups_in = [net.head_out, 512, 256, 128]
    ups_out = [512, 256, 128, 64]
    cat = list(_CAT[net.version])
    for i in range(4):
        out[f"up_convs.{i}.weight"] = (ups_in[i], ups_out[i], 2, 2)
        out[f"up_convs.{i}.bias"] = (ups_out[i],)
    for i in range(4):
        p = f"double_convs.{i}"
        out[p + ".0.weight"] = (ups_out[i], cat[i], 3, 3); out[p + ".0.bias"] = (ups_out[i],)
        bn(p + ".1", ups_out[i])
        out[p + ".3.weight"] = (ups_out[i], ups_out[i], 3, 3); out[p + ".3.bias"] = (ups_out[i],)
        bn(p + ".4", ups_out[i])
    out["input_up_conv.weight"] = (64, 32, 2, 2); out["input_up_conv.bias"] = (32,)
    p = "input_double_conv"
    out[p + ".0.weight"] = (32, 32 + net.in_channels, 3, 3); out[p + ".0.bias"] = (32,)
    bn(p + ".1", 32)
    out[p + ".3.weight"] = (32, 32, 3, 3); out[p + ".3.bias"] = (32,)
    bn(p + ".4", 32)
    out["out_conv1x1.weight"] = (net.num_classes, 32, 1, 1); out["out_conv1x1.bias"] = (net.num_classes,)
    return out

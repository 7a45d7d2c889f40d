"""TEST INFRASTRUCTURE ONLY — CPU oracle: the transformer block the reference takes from timm.

PARITY UNPINNED at this boundary.  The reference does
    from timm.models.layers import to_2tuple                  (prithvi.py:18, used :102-103)
    from timm.models.vision_transformer import Block          (prithvi.py:19; built :162-164,
                                                               :178-183; run :301-302, :321-322)
and `timm>=0.9.12` (requirements.txt:3, floor only, no lock file) is third-party, not vendored
under /root/reference and not installed here.  This file restates the published algorithm of
timm 0.9.12's ``vision_transformer.Block`` for the arguments the reference passes
(dim, num_heads, mlp_ratio, qkv_bias=True, norm_layer=nn.LayerNorm; everything else default):

    x = x + proj(softmax(q k^T * hd^-0.5) v)     q,k,v = split(qkv(LN1(x)))  packed [3, h, hd]
    x = x + fc2(GELU_erf(fc1(LN2(x))))           LayerNorm eps 1e-5; no dropout/LayerScale/DropPath

Parameter names match timm's so reference state-dicts load: norm1, attn.qkv, attn.proj, norm2,
mlp.fc1, mlp.fc2.  Cross-checked in tests against torch.nn.MultiheadAttention (same packed
in-proj layout) and F.scaled_dot_product_attention — independent, but not the reference's code.
"""
from __future__ import annotations

import math

import torch
import torch.nn as nn
import torch.nn.functional as F


def to_2tuple(x):
    return tuple(x) if isinstance(x, (tuple, list)) else (x, x)


class Attention(nn.Module):
    def __init__(self, dim: int, num_heads: int, qkv_bias: bool = True):
        super().__init__()
        assert dim % num_heads == 0
        self.num_heads = num_heads
        self.head_dim = dim // num_heads
        self.qkv = nn.Linear(dim, dim * 3, bias=qkv_bias)
        self.proj = nn.Linear(dim, dim)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        B, N, C = x.shape
        qkv = self.qkv(x).reshape(B, N, 3, self.num_heads, self.head_dim).permute(2, 0, 3, 1, 4)
        q, k, v = qkv[0], qkv[1], qkv[2]  # each [B, h, N, hd]
        att = (q @ k.transpose(-2, -1)) * (1.0 / math.sqrt(self.head_dim))
        att = torch.softmax(att, dim=-1)
        x = (att @ v).transpose(1, 2).reshape(B, N, C)
        return self.proj(x)


class Mlp(nn.Module):
    def __init__(self, dim: int, hidden: int):
        super().__init__()
        self.fc1 = nn.Linear(dim, hidden)
        self.fc2 = nn.Linear(hidden, dim)

    def forward(self, x):
        return self.fc2(F.gelu(self.fc1(x)))  # exact erf GELU


class Block(nn.Module):
    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, norm_layer=nn.LayerNorm, **_):
        super().__init__()
        self.norm1 = norm_layer(dim)
        self.attn = Attention(dim, num_heads, qkv_bias)
        self.norm2 = norm_layer(dim)
        self.mlp = Mlp(dim, int(dim * mlp_ratio))

    def forward(self, x):
        x = x + self.attn(self.norm1(x))
        x = x + self.mlp(self.norm2(x))
        return x

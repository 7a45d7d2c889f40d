"""Host-side helpers on the hot path's boundary (reference: /root/reference/src/utils.py)."""
from __future__ import annotations

import torch
from torch import nn


def initialize_classification_layer_bias(layer: nn.Linear | nn.Conv2d, class_distribution: list[float]) -> None:
    """Classifier bias = log class prior (reference utils.py:174-188); 2 classes: log(p1/p0) fill."""
    dist = torch.tensor(class_distribution, dtype=torch.float32) + 1e-6
    assert torch.isclose(dist.sum(), torch.tensor(1.0)), f"Must sum to 1, got distribution: {class_distribution}"
    assert len(class_distribution) > 1, "Class distribution must have at least 2 classes"
    with torch.no_grad():
        if len(dist) == 2:
            layer.bias.fill_((dist[1] / dist[0]).log())
        else:
            layer.bias.copy_(dist.log())

"""Round-4 fixtures from the *imported reference* (build container only - /root/reference never travels).

    python tests/golden/make_golden_r4.py

class_bias_prior.npz: what the reference's `initialize_classification_layer_bias(layer, class_distribution)`
(/root/reference/src/utils.py:174-188) leaves in the bias of the classification layer, for the two branches the function has:
  * two classes: every bias entry = log(p1 / p0)  ([0.7, 0.3] and the reverse, on a Conv2d with 2 and with 1 output channel);
  * more classes: bias = log(p + 1e-6)  (uniform, the non-uniform [0.5, 0.3, 0.15, 0.05], a 10-class ramp), on Conv2d and Linear.
  * the function's own sum check runs on the eps-shifted distribution in float32, so a valid 10-class ramp is REFUSED (10 x 1e-6
    pushes the sum past isclose's tolerance): the drop-in must refuse the same inputs (`raised` = 1, bias left as constructed).
Stored per case: the distribution, whether the reference raised its AssertionError, and the resulting bias vector (float32).
"""
from __future__ import annotations

import sys
from pathlib import Path

import numpy as np
import torch
from torch import nn

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
sys.path.insert(0, str(HERE.parents[1]))

import ref_harness  # noqa: E402

CASES = {
    "two_class_conv2": ([0.7, 0.3], lambda: nn.Conv2d(8, 2, 1)),
    "two_class_rev_conv1": ([0.3, 0.7], lambda: nn.Conv2d(8, 1, 1)),
    "uniform4_conv": ([0.25, 0.25, 0.25, 0.25], lambda: nn.Conv2d(32, 4, 1)),
    "nonuniform4_conv": ([0.5, 0.3, 0.15, 0.05], lambda: nn.Conv2d(32, 4, 1)),
    "nonuniform4_linear": ([0.5, 0.3, 0.15, 0.05], lambda: nn.Linear(16, 4)),
    "ramp10_conv": ([(i + 1) / 55.0 for i in range(10)], lambda: nn.Conv2d(32, 10, 1)),
    "ramp6_linear": ([(i + 1) / 21.0 for i in range(6)], lambda: nn.Linear(16, 6)),
    "not_normalised": ([0.5, 0.3, 0.1], lambda: nn.Conv2d(8, 3, 1)),
}


def main() -> None:
    ref = ref_harness.load()
    out = {}
    for name, (dist, make) in CASES.items():
        torch.manual_seed(0)
        layer = make()
        raised = 0
        try:
            ref.utils.initialize_classification_layer_bias(layer, dist)
        except AssertionError:
            raised = 1
        out[f"{name}.raised"] = np.asarray([raised], dtype=np.int64)
        out[f"{name}.dist"] = np.asarray(dist, dtype=np.float64)
        out[f"{name}.bias"] = layer.bias.detach().numpy().astype(np.float32)
        out[f"{name}.kind"] = np.asarray([0 if isinstance(layer, nn.Conv2d) else 1, layer.bias.numel()], dtype=np.int64)
    np.savez_compressed(HERE / "class_bias_prior.npz", **out)
    print("wrote class_bias_prior.npz:", {k: v.tolist() for k, v in out.items() if k.endswith(".bias")})


if __name__ == "__main__":
    main()

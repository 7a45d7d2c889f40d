"""TEST INFRASTRUCTURE ONLY — CPU oracle for the s2lc hot path.

This package is a plain-PyTorch (CPU, fp32, eager) restatement of the reference's
segmentation / MAE forward+backward path.  It exists so that tests, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg can check / time the reference algorithm on machines where
`/root/reference` does not exist (the GPU box).

Nothing under `sentinel2-landcover-classification_amd/` (the product) may import from here:
the product path is HIP-only and fails loudly when the HIP extension is missing.

Pinning: the reference holds no golden vectors of its own, so every module here
is pinned against outputs of the reference itself, generated in the build container by
`tests/golden/make_golden.py` (which imports `/root/reference/src`) and committed as small
fixtures under `tests/golden/*.npz`.  The one exception is the transformer `Block`
(`oracle/vit_block_ref.py`): its arithmetic lives in third-party `timm>=0.9.12`
(requirements.txt:3), which is neither vendored nor installed — **parity unpinned** at that
boundary; it is cross-checked against torch's own CPU SDPA / LayerNorm / GELU instead.
"""

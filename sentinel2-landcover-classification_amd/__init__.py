"""s2lc-mi355x: MI355X-native hot path of sentinel2-landcover-classification.

The directory is named after the reference repo (`sentinel2-landcover-classification_amd`), which
is not a Python identifier; import it through the `s2lc_amd` shim at the repository root.
"""
__all__ = ["modules", "plan", "losses", "engine"]

from .layers import ScatLayer, ScatLayerj2   # noqa: F401

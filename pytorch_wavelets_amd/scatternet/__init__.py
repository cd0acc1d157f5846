from .layers import ScatLayer   # noqa: F401

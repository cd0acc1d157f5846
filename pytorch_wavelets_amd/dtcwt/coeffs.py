"""Filter-table access with the reference's names (pytorch_wavelets/dtcwt/coeffs.py:34-117)."""
from ..filters import biort, level1, qshift   # noqa: F401

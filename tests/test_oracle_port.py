"""The C/OpenMP port used as CPU baseline must agree with the numpy oracle (which is pinned against the reference)."""
import numpy as np
import pytest

from oracle import dwt_port, wavelet_oracle as wo
from pytorch_wavelets_amd import filters as F


@pytest.mark.parametrize('wave,mode,J,shape', [
    ('db4', 'symmetric', 3, (2, 3, 64, 80)), ('db2', 'zero', 2, (1, 2, 37, 50)), ('db3', 'reflect', 2, (1, 1, 40, 33)),
    ('db8', 'periodization', 3, (1, 2, 128, 96)), ('db4', 'periodic', 2, (1, 1, 45, 52)), ('haar', 'zero', 1, (1, 3, 64, 64)),
    ('db3', 'periodization', 2, (1, 1, 63, 50)),
])
def test_port_matches_oracle(wave, mode, J, shape):
    rng = np.random.RandomState(0)
    x = rng.randn(*shape).astype(np.float32)
    h0, h1 = F.dwt_analysis_taps(wave)
    g0, g1 = F.dwt_synthesis_taps(wave)
    oyl, oyh = wo.dwt_forward(x.astype(np.float64), J, h0, h1, h0, h1, mode)
    c = dwt_port.forward(x, J, h0, h1, mode, threads=2)
    yl, yh = dwt_port.unpack(c, shape, J, len(h0), mode)
    assert np.abs(yl - oyl).max() / np.abs(oyl).max() < 1e-5
    for a, b in zip(yh, oyh):
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-5
    rec = dwt_port.inverse(c, shape, J, g0, g1, mode, threads=2)
    orec = wo.dwt_inverse(oyl, oyh, g0, g1, g0, g1, mode)
    orec = orec[..., :shape[2], :shape[3]]
    assert np.abs(rec - orec).max() / np.abs(orec).max() < 1e-5

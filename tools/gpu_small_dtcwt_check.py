"""Small planes through the modules: DTCWTForward / DTCWTInverse J = 1..3, ScatLayer (inference, training, odd sizes, colour
combination) and ScatLayerj2 with the small-plane level-1 kernel against the same modules on the tile kernels (no_stream)."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import _lib
dev = 'cuda:0'
lib = _lib.get()
bad = n = 0
small = 0
torch.manual_seed(0)
for shape in ((64, 3, 32, 32), (40, 2, 16, 24), (33, 1, 31, 32), (128, 3, 8, 8), (20, 3, 36, 28), (17, 3, 32, 32)):
    x = torch.randn(*shape, device=dev)
    mods = [('dtcwt J=%d' % J, pw.DTCWTForward(J=J).to(dev)) for J in (1, 2, 3)] + \
           [('scat', pw.ScatLayer().to(dev)), ('scat zero', pw.ScatLayer(mode='zero').to(dev)), ('scat near_sym_b', pw.ScatLayer(biort='near_sym_b').to(dev))]
    if shape[1] == 3:
        mods.append(('scat colour', pw.ScatLayer(combine_colour=True).to(dev)))
    if min(shape[2:]) >= 16:
        mods.append(('scatj2', pw.ScatLayerj2().to(dev)))
    for name, m in mods:
        outs = {}
        for ns in (0, 1):
            lib.wl_set_option(b'no_stream', ns)
            xg = x.clone().requires_grad_(True)
            c0 = pw.launch_count()
            y = m(xg)
            ks = pw.kernels_since(c0)
            if ns == 0 and any('Small' in k for k in ks):
                small += 1
            flat = [y] if torch.is_tensor(y) else [y[0]] + list(y[1])
            g, = torch.autograd.grad(sum((t * t).sum() for t in flat), xg)
            outs[ns] = [t.detach() for t in flat] + [g]
        lib.wl_set_option(b'no_stream', 0)
        for a, b in zip(outs[0], outs[1]):
            n += 1
            e = float((a - b).abs().max() / max(1.0, float(b.abs().max())))
            if a.shape != b.shape or not e < 2e-5:
                bad += 1; print('BAD', shape, name, e)
print(json.dumps({'comparisons': n, 'module_runs_on_small_plane_kernels': small, 'bad': bad}))

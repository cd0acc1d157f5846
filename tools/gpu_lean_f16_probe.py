"""Lean streaming kernels against the tile kernels (wl_set_option no_stream) on planes of 128 / 256 columns, float32 and float16:
ScatLayer, DTCWTForward J = 1 and J = 2."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
dev = 'cuda:0'; sync = torch.cuda.synchronize
for dtype in (torch.float32, torch.float16):
    for shape in ((256, 3, 224, 224), (256, 3, 160, 160), (256, 3, 192, 200), (256, 3, 128, 128)):
        x = torch.randn(*shape, device=dev).to(dtype)
        mods = {'scat': pw.ScatLayer().to(dev).to(dtype), 'dtcwt1': pw.DTCWTForward(J=1).to(dev).to(dtype), 'dtcwt2': pw.DTCWTForward(J=2).to(dev).to(dtype)}
        line = []
        with torch.no_grad():
            for name, m in mods.items():
                r = []
                for ns in (0, 1):
                    ops.set_option('no_stream', ns)
                    m(x); c0 = pw.launch_count(); m(x); ks = pw.kernels_since(c0)
                    r.append((bench.time_seq_fn(lambda: m(x), 20, sync), ks[0].split('<')[0] + ks[0][-9:]))
                ops.set_option('no_stream', 0)
                line.append('%s %.4f %s | tile %.4f' % (name, r[0][0], r[0][1], r[1][0]))
        # inverse J = 1 / J = 2 and the ScatLayer training step
        for J in (1, 2):
            with torch.no_grad():
                c = pw.DTCWTForward(J=J).to(dev).to(dtype)(x)
                ifm = pw.DTCWTInverse().to(dev).to(dtype)
                r = []
                for ns in (0, 1):
                    ops.set_option('no_stream', ns)
                    ifm(c); c0 = pw.launch_count(); ifm(c); ks = pw.kernels_since(c0)
                    r.append((bench.time_seq_fn(lambda: ifm(c), 20, sync), ks[-1].split('<')[0]))
                ops.set_option('no_stream', 0)
                line.append('inv%d %.4f %s | tile %.4f' % (J, r[0][0], r[0][1], r[1][0]))
        sl = mods['scat']

        def step():
            xg = x.detach().requires_grad_(True)
            z = sl(xg)
            return torch.autograd.grad(z, xg, z)
        r = []
        for ns in (0, 1):
            ops.set_option('no_stream', ns)
            step(); r.append(bench.time_seq_fn(step, 10, sync))
        ops.set_option('no_stream', 0)
        line.append('scat fwd+bwd %.4f | tile %.4f' % (r[0], r[1]))
        print(str(dtype)[6:], shape, ' ;  '.join(line), flush=True)

"""Same-box A/B of builds of the strip translation unit on config 5's level 1 (32x16x2048x2048 float16, db8 periodization):
forward (WlAfbStrip) and inverse (WlSfbStrip) time per launch.  Each library runs in its own process.
usage: python tools/gpu_cfg5_ab.py [lib.so ...]     ('' = the product build)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if os.environ.get('_CFG5_AB_CHILD'):
    sys.path.insert(0, ROOT)
    import torch
    import pytorch_wavelets_amd as pw
    from pytorch_wavelets_amd import ops
    dev = 'cuda:0'
    a, b = torch.empty(64 << 20, device=dev), torch.empty(64 << 20, device=dev)

    def timeit(fn, n=10):
        for _ in range(4):
            fn()
        torch.cuda.synchronize()
        res = []
        for _ in range(3):
            for _ in range(20):
                b.copy_(a)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                fn()
            e1.record()
            torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1) / n)
        return sorted(res)[1]
    W = int(os.environ.get('_CFG5_AB_W', '2048'))
    xfm = pw.DWTForward(J=1, wave='db8', mode='periodization').to(dev).half()
    ifm = pw.DWTInverse(wave='db8', mode='periodization').to(dev).half()
    x = torch.randn(32, 16, W, W, device=dev).half()
    with torch.no_grad():
        ops.STREAM_FORCE = True
        c = xfm(x)
        tf = timeit(lambda: xfm(x)); kf = pw.last_kernel()
        ti = timeit(lambda: ifm(c)); ki = pw.last_kernel()
        rt = ifm(c)
        err = float((rt.float() - x.float()).abs().max())
        chk = float(c[0].float().abs().mean())
    print(json.dumps({'lib': os.environ.get('WL_LIB', ''), 'W': W, 'roundtrip_abs_err': round(err, 5), 'yl_absmean': round(chk, 6), 'fwd_ms': round(tf, 4), 'inv_ms': round(ti, 4), 'fwd_kernel': kf, 'inv_kernel': ki}), flush=True)
    sys.exit(0)

libs = sys.argv[1:] or ['']
for rep in range(2):
    for lib in libs:
        env = dict(os.environ, _CFG5_AB_CHILD='1')
        if lib:
            env['WL_LIB'] = os.path.join(ROOT, lib)
        else:
            env.pop('WL_LIB', None)
        r = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
        print([l for l in r.stdout.decode().splitlines() if l.startswith('{')][-1:] or r.stdout.decode()[-400:], flush=True)

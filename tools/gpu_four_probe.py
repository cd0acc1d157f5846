"""Four planes per workgroup on planes narrower than 96 columns (A/B build -DWL_FOUR_MIN=64): ScatLayer / DTCWT level 1."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
dev = 'cuda:0'; sync = torch.cuda.synchronize
for shape in ((1024, 3, 64, 64), (256, 16, 64, 64), (512, 3, 80, 80), (512, 3, 72, 72), (512, 3, 88, 88)):
    x = torch.randn(*shape, device=dev)
    line = []
    with torch.no_grad():
        for name, m in (('scat', pw.ScatLayer().to(dev)), ('dtcwt1', pw.DTCWTForward(J=1).to(dev))):
            m(x); c0 = pw.launch_count(); m(x); ks = pw.kernels_since(c0)
            line.append('%s %.4f %s' % (name, bench.time_seq_fn(lambda: m(x), 20, sync), ks[0].split('<')[0] + ks[0][-9:]))
    print(os.environ.get('WL_LIB'), shape, ' ; '.join(line), flush=True)

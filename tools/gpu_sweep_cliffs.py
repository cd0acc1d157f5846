"""A sweep over the module API looking for performance cliffs: modes, depths, odd sizes, dtypes, filter tables.  One line per case:
forward / inverse time, fraction of the HBM roofline at the algorithmic bytes, kernels."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
dev = 'cuda:0'; sync = torch.cuda.synchronize


def run(tag, fx, fi, x):
    with torch.no_grad():
        c = fx(x)
        c0 = pw.launch_count(); fx(x); kf = pw.kernels_since(c0)
        c0 = pw.launch_count(); fi(c); ki = pw.kernels_since(c0)
        tf, ti = bench.time_seq_fn(lambda: fx(x), 20, sync), bench.time_seq_fn(lambda: fi(c), 20, sync)
    yl, yh = c
    b = x.element_size() * (x.numel() + yl.numel() + sum(h.numel() for h in yh))
    short = lambda ks: ','.join(sorted(set(k.split('<')[0] for k in ks)))
    print('%-46s fwd %.4f ms %.3f [%s]   inv %.4f ms %.3f [%s]' % (tag, tf, b / tf / 8e9, short(kf), ti, b / ti / 8e9, short(ki)), flush=True)


x = torch.randn(128, 3, 512, 512, device=dev)
for mode in ('zero', 'symmetric', 'reflect', 'periodization', 'periodic'):
    run('dwt db4 J=3 %s 128x3x512^2' % mode, pw.DWTForward(J=3, wave='db4', mode=mode).to(dev), pw.DWTInverse(wave='db4', mode=mode).to(dev), x)
for J in (1, 2, 4, 5):
    run('dwt db4 J=%d symmetric 128x3x512^2' % J, pw.DWTForward(J=J, wave='db4', mode='symmetric').to(dev), pw.DWTInverse(wave='db4', mode='symmetric').to(dev), x)
for wave in ('haar', 'bior2.2', 'bior4.4', 'sym8', 'coif2'):
    run('dwt %s J=3 symmetric 128x3x512^2' % wave, pw.DWTForward(J=3, wave=wave, mode='symmetric').to(dev), pw.DWTInverse(wave=wave, mode='symmetric').to(dev), x)
run('dwt db4 J=3 symmetric fp16 128x3x512^2', pw.DWTForward(J=3, wave='db4', mode='symmetric').to(dev).half(), pw.DWTInverse(wave='db4', mode='symmetric').to(dev).half(), x.half())
for hw in ((500, 500), (511, 513), (224, 224), (384, 640)):
    xs = torch.randn(128, 3, *hw, device=dev)
    run('dwt db4 J=3 symmetric 128x3x%dx%d' % hw, pw.DWTForward(J=3, wave='db4', mode='symmetric').to(dev), pw.DWTInverse(wave='db4', mode='symmetric').to(dev), xs)
del x
x = torch.randn(64, 3, 512, 512, device=dev)
for biort, qshift in (('near_sym_a', 'qshift_a'), ('near_sym_b', 'qshift_b'), ('antonini', 'qshift_c'), ('legall', 'qshift_06'), ('near_sym_b', 'qshift_d')):
    run('dtcwt %s/%s J=3 64x3x512^2' % (biort, qshift), pw.DTCWTForward(J=3, biort=biort, qshift=qshift).to(dev), pw.DTCWTInverse(biort=biort, qshift=qshift).to(dev), x)
for J in (1, 2, 4):
    run('dtcwt near_sym_a/qshift_a J=%d 64x3x512^2' % J, pw.DTCWTForward(J=J).to(dev), pw.DTCWTInverse().to(dev), x)
run('dtcwt J=3 fp16 64x3x512^2', pw.DTCWTForward(J=3).to(dev).half(), pw.DTCWTInverse().to(dev).half(), x.half())
for hw in ((500, 500), (224, 224), (510, 514)):
    xs = torch.randn(64, 3, *hw, device=dev)
    run('dtcwt J=3 64x3x%dx%d' % hw, pw.DTCWTForward(J=3).to(dev), pw.DTCWTInverse().to(dev), xs)

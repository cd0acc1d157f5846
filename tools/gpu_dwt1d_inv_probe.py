"""DWT1DInverse (one wl_synth1d launch per level) on the shape of the f3 bench line."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
dev = 'cuda:0'
x = torch.randn(64, 16, 65536, device=dev)
for J in (1, 3):
    f, i = pw.DWT1DForward(J=J, wave='db4', mode='symmetric').to(dev), pw.DWT1DInverse(wave='db4', mode='symmetric').to(dev)
    with torch.no_grad():
        c = f(x)
        c0 = pw.launch_count(); i(c); k = pw.kernels_since(c0)
        tf = bench.time_seq_fn(lambda: f(x), 20, torch.cuda.synchronize)
        ti = bench.time_seq_fn(lambda: i(c), 20, torch.cuda.synchronize)
    print('J=%d fwd %.4f ms (%.3f)  inv %.4f ms (%.3f of 8 TB/s at 8 B/sample) %s' % (J, tf, 8 * x.numel() / tf / 8e9, ti, 8 * x.numel() / ti / 8e9, k), flush=True)

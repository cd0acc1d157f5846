"""10- and 12-tap wavelets on the streaming analysis kernel with one set of tap pairs (WlAfbRows<.., SAME = 1>) against two sets
(the hint switched off): DWTForward J = 3 on 128x3x512x512."""
import os, sys, contextlib, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
dev = 'cuda:0'; sync = torch.cuda.synchronize
x = torch.randn(128, 3, 512, 512, device=dev)
for wave, L in (('db4', 8), ('db5', 10), ('sym5', 10), ('db6', 12), ('coif2', 12)):
    m = pw.DWTForward(J=3, wave=wave, mode='symmetric').to(dev)
    b = bench.algorithmic_bytes_fwd(128, 3, 512, 512, 3, L, 4)
    with torch.no_grad():
        m(x); c0 = pw.launch_count(); m(x); ks = pw.kernels_since(c0)
        t1 = bench.time_seq_fn(lambda: m(x), 30, sync)
        real = ops.same_banks_hint
        ops.same_banks_hint = lambda flag: contextlib.nullcontext()
        try:
            m(x); c0 = pw.launch_count(); m(x); ks0 = pw.kernels_since(c0)
            t0 = bench.time_seq_fn(lambda: m(x), 30, sync)
        finally:
            ops.same_banks_hint = real
    print('%s: one set %.4f ms (%.3f) %s | two sets %.4f ms (%.3f) %s' % (wave, t1, b / t1 / 8e9, ks, t0, b / t0 / 8e9, ks0), flush=True)

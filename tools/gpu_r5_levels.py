"""Round 5: what sets the level a process runs its kernels at?  One process: time, release every cached block (torch.cuda.empty_cache: the next
tensors come from fresh hipMalloc calls), time again, ..."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
import bench
dev = 'cuda:0'; sync = torch.cuda.synchronize
d = pw.DTCWTForward(J=3).to(dev)
f = pw.DWTForward(J=3, wave='db4', mode='symmetric').to(dev)
keep = []
for trial in range(10):
    torch.cuda.empty_cache()
    if trial >= 5:      # hold on to odd-sized blocks so that the next hipMalloc calls land elsewhere
        keep.append(torch.empty((trial * 37 + 11) << 20, dtype=torch.uint8, device=dev))
    x = torch.randn(64, 3, 512, 512, device=dev); xm = torch.randn(128, 3, 512, 512, device=dev)
    with torch.no_grad():
        for _ in range(200):
            d(x)
        td = min(bench.time_seq_fn(lambda: d(x), 100, sync) for _ in range(3))
        for _ in range(200):
            f(xm)
        tf = min(bench.time_seq_fn(lambda: f(xm), 100, sync) for _ in range(3))
    print(json.dumps({'trial': trial, 'dtcwt_fwd': round(td, 4), 'dwt_fwd': round(tf, 4), 'x_ptr_GB': round(x.data_ptr() / 2**30, 3)}), flush=True)
    del x, xm

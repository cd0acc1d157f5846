"""DWT1DForward / DWT1DInverse J=1..3 at 64x16x65536 and a few other lengths: ms, fraction of the HBM roofline at 8 B per sample (float32), round trip."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pytorch_wavelets_amd as pw
dev = 'cuda:0'; sync = torch.cuda.synchronize
for shape in ((64, 16, 65536), (64, 16, 65535), (256, 8, 16384), (8, 4, 1048576)):
    for wave, J, dt in (('db4', 3, torch.float32), ('db4', 1, torch.float32), ('db2', 2, torch.float32), ('db8', 3, torch.float32), ('db4', 3, torch.float16)):
        x = torch.randn(*shape, device=dev).to(dt)
        f = pw.DWT1DForward(J=J, wave=wave, mode='symmetric').to(dev).to(dt); i = pw.DWT1DInverse(wave=wave, mode='symmetric').to(dev).to(dt)
        with torch.no_grad():
            c = f(x); r = i(c)
            tf = min(bench.time_seq_fn(lambda: f(x), 30, sync) for _ in range(3)); ti = min(bench.time_seq_fn(lambda: i(c), 30, sync) for _ in range(3))
        b = 2 * x.numel() * x.element_size()
        print(json.dumps({'shape': shape, 'wave': wave, 'J': J, 'dtype': str(dt)[6:], 'fwd_ms': round(tf, 4), 'fwd_frac': round(b / tf / 8e9, 3), 'inv_ms': round(ti, 4),
                          'inv_frac': round(b / ti / 8e9, 3), 'rt': float((r[..., :x.shape[-1]] - x).abs().max() / x.abs().max()), 'k': pw.last_kernel()}), flush=True)

"""Round 5: the shapes the late work targeted, 30 forward + 30 inverse calls each, for a rocprofv3 --kernel-trace --stats run
(tools/gpu_r5_shapes_trace.sh) - which kernel, which grid, how long, per shape."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import pytorch_wavelets_amd as pw
dev = 'cuda:0'
CASES = [((128, 3, 224, 224), 1, 'db4', 'symmetric', torch.float32), ((128, 3, 224, 224), 3, 'db4', 'symmetric', torch.float32),
         ((512, 3, 224, 224), 3, 'db4', 'symmetric', torch.float32), ((128, 3, 640, 640), 3, 'db4', 'symmetric', torch.float32),
         ((64, 3, 1024, 1024), 3, 'db4', 'symmetric', torch.float32), ((32, 16, 512, 512), 1, 'db8', 'periodization', torch.float16),
         ((128, 16, 256, 256), 1, 'db8', 'periodization', torch.float16)]
for shape, J, wave, mode, dt in CASES:
    x = torch.randn(*shape, device=dev).to(dt)
    f = pw.DWTForward(J=J, wave=wave, mode=mode).to(dev).to(dt); i = pw.DWTInverse(wave=wave, mode=mode).to(dev).to(dt)
    with torch.no_grad():
        c = f(x)
        for _ in range(30):
            f(x)
        for _ in range(30):
            i(c)
    torch.cuda.synchronize()
    del x, c

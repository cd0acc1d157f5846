"""Five forward + inverse transforms in periodization at the metric's shape (128x3x512x512 float32 J=3 db4): the command the round-6 counter
passes run around (tools/gpu_pmc_cmd.sh r06_per "tcc1 tcc2" -- python tools/gpu_r6_per_step.py) - HBM traffic of the two fused launches."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
dev = 'cuda:0'
x = torch.randn(128, 3, 512, 512, device=dev)
f = pw.DWTForward(J=3, wave='db4', mode='periodization').to(dev); i = pw.DWTInverse(wave='db4', mode='periodization').to(dev)
with torch.no_grad():
    for _ in range(5):
        c = f(x); r = i(c)
torch.cuda.synchronize()
print('rt', float((r - x).abs().max()))

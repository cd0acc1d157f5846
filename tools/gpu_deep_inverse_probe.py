import os, sys, torch
sys.path.insert(0, '/root/repo')
import bench, pytorch_wavelets_amd as pw
dev='cuda:0'; sync=torch.cuda.synchronize
x=torch.randn(128,3,512,512,device=dev)
for J in (3,4,5,6):
    fx, fi = pw.DWTForward(J=J,wave='db4',mode='symmetric').to(dev), pw.DWTInverse(wave='db4',mode='symmetric').to(dev)
    with torch.no_grad():
        c=fx(x); c0=pw.launch_count(); r=fi(c); k=pw.kernels_since(c0)
        ti=bench.time_seq_fn(lambda: fi(c),20,sync)
    xg=x.clone().requires_grad_(True); yl,yh=fx(xg)
    c0=pw.launch_count(); g,=torch.autograd.grad(yl.sum()+sum(h.sum() for h in yh), xg); kb=pw.kernels_since(c0)
    print('J=%d inv %.4f ms %s  rt err %.2e | backward kernels %s'%(J,ti,k,float((r-x).abs().max()),kb),flush=True)

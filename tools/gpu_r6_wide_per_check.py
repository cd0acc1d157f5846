import os, sys
sys.path.insert(0, os.getcwd())
import torch, numpy as np
import pytorch_wavelets_amd as pw
from pytorch_wavelets_amd import ops
dev='cuda:0'
bad=0
def err(p,q): return float((p.float()-q.float()).abs().max()/max(1e-6,float(q.float().abs().max())))
for (shape,wave,J,dt) in [((64,3,1024,1024),'db4',3,torch.float32),((64,3,1024,1024),'db2',4,torch.float32),((32,3,2048,2048),'db4',4,torch.float32),((64,3,1024,1024),'db5',3,torch.float16),
                          ((96,3,1024,768),'db6',3,torch.float32),((100,3,520,1032),'db3',3,torch.float32),((32,16,2048,2048),'db8',4,torch.float16),((128,3,768,768),'db8',3,torch.float32),
                          ((40,8,1024,1024),'haar',3,torch.float16),((64,3,1536,1536),'db4',3,torch.float32)]:
    x=torch.randn(*shape,device=dev).to(dt)
    f=pw.DWTForward(J=J,wave=wave,mode='periodization').to(dev).to(dt); i=pw.DWTInverse(wave=wave,mode='periodization').to(dev).to(dt)
    out={}
    for flag in (False,True):
        ops.ROWS_PER=ops.IROWS_PER=flag; ops._FUSED_DECLINED.clear()
        with torch.no_grad():
            c0=pw.launch_count(); yl,yh=f(x); kf=[k for k in pw.kernels_since(c0) if not k.endswith(')')]
            c0=pw.launch_count(); r=i((yl,yh)); ki=[k for k in pw.kernels_since(c0) if not k.endswith(')')]
        out[flag]=(yl,yh,r,kf,ki)
    a,b=out[False],out[True]
    es=[err(b[0],a[0])]+[err(p,q) for p,q in zip(b[1],a[1])]+[err(b[2],a[2]), err(b[2],x)]
    tol=4e-3 if dt==torch.float16 else 1e-5
    ok=max(es[:-1])<tol
    bad+=not ok
    print(shape,wave,J,str(dt)[6:],'max err %.1e rt %.1e'%(max(es[:-1]),es[-1]),'OK' if ok else 'BAD',[k[:24] for k in b[3]],[k[:24] for k in b[4]])
print('bad',bad)

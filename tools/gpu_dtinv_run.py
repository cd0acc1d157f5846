"""A few launches of DTCWTInverse J=3 on 64x3x512x512 for counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
dev = torch.device('cuda:0')
xfm, ifm = pw.DTCWTForward(J=3).to(dev), pw.DTCWTInverse().to(dev)
x = torch.randn(64, 3, 512, 512, device=dev)
with torch.no_grad():
    c = xfm(x)
    for _ in range(4):
        ifm(c)
torch.cuda.synchronize()

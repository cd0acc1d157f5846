"""A few launches of DTCWTForward (J from argv, default 2) on 64x3x512x512 for counter passes."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pytorch_wavelets_amd as pw
J = int(sys.argv[1]) if len(sys.argv) > 1 else 2
dev = torch.device('cuda:0')
m = pw.DTCWTForward(J=J).to(dev)
x = torch.randn(64, 3, 512, 512, device=dev)
with torch.no_grad():
    for _ in range(4):
        m(x)
torch.cuda.synchronize()

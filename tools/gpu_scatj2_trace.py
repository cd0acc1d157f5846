"""Kernel trace target: ScatLayerj2 forward (inference) on 64x3x256x256, 20 calls.  usage (on the GPU box):
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <out> -o j2 -- python tools/gpu_scatj2_trace.py"""
import os
import sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_wavelets_amd as pw
x = torch.randn(64, 3, 256, 256, device='cuda:0')
m = pw.ScatLayerj2().to('cuda:0')
with torch.no_grad():
    for _ in range(20):
        m(x)
torch.cuda.synchronize()

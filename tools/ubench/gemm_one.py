"""One GEMM shape, a few launches, for rocprofv3 counter passes: python tools/ubench/gemm_one.py [pre]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd import ops
R = 36192
A = torch.randn(R, 1024, device='cuda'); B = torch.randn(1024, 2048, device='cuda'); C = torch.empty(R, 2048, device='cuda')
if 'pre' in sys.argv:
    A, B = ops.split3(A), ops.split3(B)
for _ in range(4):
    ops.gemm(A, B, C)
torch.cuda.synchronize()

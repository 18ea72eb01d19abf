"""Times asr_gemm_f32 on the cfg3 shapes (diagnostic).  python tools/ubench/gemm_bench.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd import ops

R = 36192
cases = [('NN xproj   ', (R, 1024), (1024, 2048), False, False),
         ('NN xproj0  ', (R, 123), (123, 2048), False, False),
         ('NN logits  ', (R, 1024), (1024, 3436), False, False),
         ('NN xhproj  ', (R, 1024), (1024, 512), False, False),
         ('NT dX      ', (R, 2048), (1024, 2048), False, True),
         ('NT dXh     ', (R, 512), (1024, 512), False, True),
         ('NT dH      ', (R, 3436), (1024, 3436), False, True),
         ('TN dWx     ', (R, 1024), (R, 2048), True, False),
         ('TN dWh     ', (R, 512), (R, 2048), True, False),
         ('TN dM      ', (R, 128), (R, 6144), True, False),
         ('TN dWout   ', (R, 1024), (R, 3436), True, False)]
for name, sa, sb, ta, tb in cases:
    A = torch.randn(sa, device='cuda'); B = torch.randn(sb, device='cuda')
    M = sa[1] if ta else sa[0]; K = sa[0] if ta else sa[1]; N = sb[0] if tb else sb[1]
    C = torch.empty(M, N, device='cuda')
    for _ in range(2):
        ops.gemm(A, B, C, ta=ta, tb=tb)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 10
    for _ in range(n):
        ops.gemm(A, B, C, ta=ta, tb=tb)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / n
    print('%s M=%6d N=%5d K=%6d  %8.1f us  %6.1f TFLOP/s' % (name, M, N, K, dt * 1e6, 2.0 * M * N * K / dt / 1e12), flush=True)

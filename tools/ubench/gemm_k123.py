import os, sys, time, torch
sys.path.insert(0, '/root/repo')
from asr_amd import ops
R = 36192
cases = [('NN xproj0 K=123', (R, 123), (123, 2048), False, False),
         ('NN xhproj0 K=123', (R, 123), (123, 512), False, False),
         ('TN dWx0 M=123', (R, 123), (R, 2048), True, False),
         ('TN dWxh0 M=123', (R, 123), (R, 512), True, False),
         ('NN xproj0 K=128', (R, 128), (128, 2048), False, False),
         ('TN dWx0 M=128', (R, 128), (R, 2048), True, False)]
for prec in ('f32x3', 'f32'):
    ops.PRECISION = prec
    for name, sa, sb, ta, tb in cases:
        A = torch.randn(sa, device='cuda'); B = torch.randn(sb, device='cuda')
        M = sa[1] if ta else sa[0]; K = sa[0] if ta else sa[1]; N = sb[0] if tb else sb[1]
        C = torch.empty(M, N, device='cuda')
        for _ in range(2): ops.gemm(A, B, C, ta=ta, tb=tb)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): ops.gemm(A, B, C, ta=ta, tb=tb)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        print('%s %-18s %8.1f us %6.1f TFLOP/s' % (prec, name, dt * 1e6, 2.0 * M * N * K / dt / 1e12), flush=True)

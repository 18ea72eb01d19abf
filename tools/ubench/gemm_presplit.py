"""Speed of the split GEMM with operands split once per tensor (ops.split3 + asr_gemm_presplit_ws) against the
on-the-fly split (asr_gemm_bf16x3_ws, 6 products) on the cfg3 shapes, and the cost of the split pass itself.
Bit-identity of the two is tests/test_gpu_kernels.py::test_gemm_presplit_bit_identical.
python tools/ubench/gemm_presplit.py"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd import ops


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


R = 36192
cases = [('NN xproj   ', (R, 1024), (1024, 2048), False, False),
         ('NN logits  ', (R, 1024), (1024, 3436), False, False),
         ('NN xhproj  ', (R, 1024), (1024, 512), False, False),
         ('NT dX      ', (R, 2048), (1024, 2048), False, True),
         ('NT dH      ', (R, 3436), (1024, 3436), False, True),
         ('TN dWx     ', (R, 1024), (R, 2048), True, False),
         ('TN dWh     ', (R, 512), (R, 2048), True, False),
         ('TN dWout   ', (R, 1024), (R, 3436), True, False)]
for name, sa, sb, ta, tb in cases:
    A = torch.randn(sa, device='cuda'); B = torch.randn(sb, device='cuda')
    M = sa[1] if ta else sa[0]; K = sa[0] if ta else sa[1]; N = sb[0] if tb else sb[1]
    C = torch.empty(M, N, device='cuda'); C2 = torch.empty(M, N, device='cuda')
    As, Bs = ops.split3(A), ops.split3(B)
    t_fly = timed(lambda: ops.gemm(A, B, C, ta, tb))
    t_pre = timed(lambda: ops.gemm(As, Bs, C2, ta, tb))
    t_pa = timed(lambda: ops.gemm(As, B, C2, ta, tb))
    t_pb = timed(lambda: ops.gemm(A, Bs, C2, ta, tb))
    ops.gemm(As, Bs, C2, ta, tb)
    t_sa = timed(lambda: ops.split3(A, As.planes.view(-1)))
    t_sb = timed(lambda: ops.split3(B, Bs.planes.view(-1)))
    fl = 2.0 * M * N * K
    print('%s M=%6d N=%5d K=%6d  on the fly %7.1f us (%5.1f TF)  both pre-split %7.1f us (%5.1f TF)  A only %7.1f  '
          'B only %7.1f  | split A %6.1f us  split B %6.1f us  same bits: %s'
          % (name, M, N, K, t_fly, fl / t_fly * 1e-6, t_pre, fl / t_pre * 1e-6, t_pa, t_pb, t_sa, t_sb,
             bool(torch.equal(C, C2))), flush=True)

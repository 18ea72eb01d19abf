"""Same (M, N, K) in the NN / NT / TN operand layouts: isolates the cost of the staging path (diagnostic)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd import ops

for (M, N, K) in [(36192, 1024, 2048), (36192, 2048, 1024), (4096, 4096, 4096)]:
    for name, ta, tb in (('NN', False, False), ('NT', False, True), ('TN', True, False), ('TT', True, True)):
        A = torch.randn((K, M) if ta else (M, K), device='cuda')
        B = torch.randn((N, K) if tb else (K, N), device='cuda')
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

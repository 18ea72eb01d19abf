"""Accuracy and speed of the split-bf16 GEMM against the exact-fp32 MFMA GEMM (diagnostic).
python tools/ubench/gemm_split_check.py"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd import ops, _lib

L = _lib.lib()
ws = torch.empty(32 << 20, dtype=torch.float32, device='cuda')


def run(mode, A, B, C, ta, tb, M, N, K):
    s = torch.cuda.current_stream().cuda_stream
    a = (int(ta), int(tb), M, N, K, 1.0, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), 0.0, C.data_ptr(),
         C.stride(0), None, ws.data_ptr(), ws.numel() * 4, s)
    rc = L.asr_gemm_f32_ws(*a) if mode == 0 else L.asr_gemm_bf16_ws(*a) if mode == -1 else L.asr_gemm_bf16x3_ws(mode, *a)
    assert rc == 0, rc


# accuracy vs float64 on awkward shapes (edges, K tails, all transposes)
rng = np.random.default_rng(0)
for (M, N, K) in [(200, 130, 70), (257, 384, 1000), (128, 128, 16), (1000, 515, 123)]:
    for ta in (False, True):
        for tb in (False, True):
            a = (rng.standard_normal((K, M) if ta else (M, K)) * np.exp(rng.standard_normal((K, M) if ta else (M, K)) * 2)).astype(np.float32)
            b = rng.standard_normal((N, K) if tb else (K, N)).astype(np.float32)
            ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
            scale = np.abs(a.T if ta else a).astype(np.float64) @ np.abs(b.T if tb else b).astype(np.float64)
            A, B = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
            C = torch.empty(M, N, device='cuda')
            out = []
            for mode in (0, 8, 6, 1, -1):
                C.fill_(float('nan'))
                run(mode, A, B, C, ta, tb, M, N, K)
                torch.cuda.synchronize()
                out.append(float(np.max(np.abs(C.cpu().numpy() - ref) / scale)))
            print('M=%4d N=%4d K=%4d ta=%d tb=%d  err/(|a||b|): f32 %.2e  x3(8) %.2e  x3(6) %.2e  bf16new %.2e bf16old %.2e'
                  % (M, N, K, ta, tb, *out), flush=True)

R = 36192
cases = [('NN xproj   ', (R, 1024), (1024, 2048), False, False),
         ('NN logits  ', (R, 1024), (1024, 3436), False, False),
         ('NN xhproj  ', (R, 1024), (1024, 512), False, False),
         ('NT dX      ', (R, 2048), (1024, 2048), False, True),
         ('NT dH      ', (R, 3436), (1024, 3436), False, True),
         ('TN dWx     ', (R, 1024), (R, 2048), True, False),
         ('TN dM      ', (R, 128), (R, 6144), True, False),
         ('TN dWout   ', (R, 1024), (R, 3436), True, False)]
for name, sa, sb, ta, tb in cases:
    A = torch.randn(sa, device='cuda'); B = torch.randn(sb, device='cuda')
    M = sa[1] if ta else sa[0]; K = sa[0] if ta else sa[1]; N = sb[0] if tb else sb[1]
    C = torch.empty(M, N, device='cuda')
    line = '%s M=%6d N=%5d K=%6d ' % (name, M, N, K)
    for mode in (0, 8, 6, 1, -1):
        for _ in range(2):
            run(mode, A, B, C, ta, tb, M, N, K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            run(mode, A, B, C, ta, tb, M, N, K)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
        line += ' | %s %7.1f us %6.1f TF' % ({0: 'f32', 8: 'x3/8', 6: 'x3/6', 1: 'bf16n', -1: 'bf16o'}[mode], dt * 1e6, 2.0 * M * N * K / dt / 1e12)
    print(line, flush=True)

"""GEMM fuzz through the C-ABI: random M, N, K (1 .. a few thousand, tile edges +-1, K tails, split-K shapes), both
transposes, padded leading dimensions on A, B and C, alpha / beta / bias on or off, for the fp32 MFMA kernel, the split
(fp32-grade, 6 and 8 products) kernel and the bf16 mode; against float64 numpy, element-wise relative to |A|.|B|; the
padding columns of C must come back untouched.  Test infrastructure.
  python tools/fuzz_gemm.py [cases] [seed]"""
import os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from asr_amd import _lib

L = _lib.lib()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
ws = torch.empty(64 << 20, device='cuda')
stream = torch.cuda.current_stream().cuda_stream
EDGE = [1, 2, 3, 5, 15, 16, 17, 31, 32, 33, 63, 64, 65, 127, 128, 129, 255, 256, 257, 383, 512, 640, 1000, 2048]
BOUND = {'f32': 1.5e-6, 'x6': 1.5e-6, 'x8': 1.5e-6, 'bf16': 1.2e-2}
bad = 0
worst = dict((k, 0.0) for k in BOUND)
t0 = time.time()
for k in range(n):
    M = int(EDGE[rng.integers(len(EDGE))]); N = int(EDGE[rng.integers(len(EDGE))])
    K = int(rng.choice([1, 2, 3, 4, 7, 8, 15, 16, 17, 31, 33, 48, 100, 123, 256, 511, 1024, 4100, 9000, 36192]))
    if M * N * K > 1.5e9:
        K = max(1, int(1.5e9 // (M * N)))
    ta, tb = int(rng.integers(2)), int(rng.integers(2))
    pa, pb, pc = int(rng.integers(0, 3)) * 4, int(rng.integers(0, 3)) * 4, int(rng.integers(0, 4))
    alpha = float(rng.choice([1.0, 0.5, -2.0])); beta = float(rng.choice([0.0, 1.0, 0.25])); use_bias = bool(rng.integers(2))
    sa = (K, M + pa) if ta else (M, K + pa); sb = (N, K + pb) if tb else (K, N + pb)
    A = (rng.standard_normal(sa) * np.exp(rng.standard_normal(sa))).astype(np.float32)
    B = (rng.standard_normal(sb) * np.exp(rng.standard_normal(sb))).astype(np.float32)
    C0 = rng.standard_normal((M, N + pc)).astype(np.float32)
    bias = rng.standard_normal(N).astype(np.float32)
    opA = (A[:, :M].T if ta else A[:, :K]).astype(np.float64); opB = (B[:, :K].T if tb else B[:, :N]).astype(np.float64)
    ref = alpha * (opA @ opB) + beta * C0[:, :N] + (bias if use_bias else 0.0)
    scale = abs(alpha) * (np.abs(opA) @ np.abs(opB)) + abs(beta) * np.abs(C0[:, :N]) + (np.abs(bias) if use_bias else 0.0) + 1e-30
    dA, dB, db = torch.from_numpy(A).cuda(), torch.from_numpy(B).cuda(), torch.from_numpy(bias).cuda()
    for kind in ('f32', 'x6', 'x8', 'bf16'):
        dC = torch.from_numpy(C0).cuda()
        args = (ta, tb, M, N, K, alpha, dA.data_ptr(), A.shape[1], dB.data_ptr(), B.shape[1], beta, dC.data_ptr(), N + pc,
                db.data_ptr() if use_bias else None, ws.data_ptr(), ws.numel() * 4, stream)
        if kind == 'f32':
            rc = L.asr_gemm_f32_ws(*args)
        elif kind == 'bf16':
            rc = L.asr_gemm_bf16_ws(*args)
        else:
            rc = L.asr_gemm_bf16x3_ws(int(kind[1]), *args)
        torch.cuda.synchronize()
        got = dC.cpu().numpy()
        e = float(np.max(np.abs(got[:, :N] - ref) / scale)) if rc == 0 else float('nan')
        tag = 'M=%d N=%d K=%d ta=%d tb=%d lda+%d ldb+%d ldc+%d alpha=%g beta=%g bias=%d %s' % (M, N, K, ta, tb, pa, pb, pc, alpha,
                                                                                            beta, use_bias, kind)
        if rc != 0 or not (e < BOUND[kind]) or not (got[:, N:] == C0[:, N:]).all() or not np.isfinite(got).all():
            bad += 1
            print('[%3d] FAILED %s rc=%d err=%.2e padding untouched: %s' % (k, tag, rc, e, bool((got[:, N:] == C0[:, N:]).all())),
                  flush=True)
        else:
            worst[kind] = max(worst[kind], e)
    if k % 50 == 49:
        print('[%3d] ... worst so far %s (%.0f s)' % (k, ' '.join('%s %.2e' % kv for kv in worst.items()), time.time() - t0), flush=True)
print('FUZZ GEMM %d cases x 4 kernels, %d FAILED; worst element-wise error relative to |A|.|B|: %s'
      % (n, bad, ' '.join('%s %.2e' % kv for kv in worst.items())))
sys.exit(1 if bad else 0)

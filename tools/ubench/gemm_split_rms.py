"""RMS error of the GEMM modes against float64 on Gaussian operands (diagnostic)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from asr_amd import _lib
L = _lib.lib()
ws = torch.empty(32 << 20, dtype=torch.float32, device='cuda')
rng = np.random.default_rng(0)
for (M, N, K, ta, tb) in [(1024, 1024, 1024, 0, 0), (1024, 1024, 2048, 0, 1), (512, 1024, 36192, 1, 0), (1024, 512, 128, 0, 0)]:
    a = rng.standard_normal((K, M) if ta else (M, K)).astype(np.float32)
    b = (rng.standard_normal((N, K) if tb else (K, N)) * 0.05).astype(np.float32)
    ref = (a.T if ta else a).astype(np.float64) @ (b.T if tb else b).astype(np.float64)
    A, B = torch.from_numpy(a).cuda(), torch.from_numpy(b).cuda()
    C = torch.empty(M, N, device='cuda')
    out = []
    for mode in (0, 8, 6, 1):
        s = torch.cuda.current_stream().cuda_stream
        args = (ta, tb, M, N, K, 1.0, A.data_ptr(), A.stride(0), B.data_ptr(), B.stride(0), 0.0, C.data_ptr(), C.stride(0), None,
                ws.data_ptr(), ws.numel() * 4, s)
        rc = L.asr_gemm_f32_ws(*args) if mode == 0 else L.asr_gemm_bf16x3_ws(mode, *args)
        assert rc == 0
        torch.cuda.synchronize()
        e = C.cpu().numpy().astype(np.float64) - ref
        out.append((np.sqrt(np.mean(e * e)) / np.sqrt(np.mean(ref * ref)), np.mean(e) / np.sqrt(np.mean(ref * ref))))
    c32 = (a.T if ta else a) @ (b.T if tb else b)
    e = c32.astype(np.float64) - ref
    out.append((np.sqrt(np.mean(e * e)) / np.sqrt(np.mean(ref * ref)), np.mean(e) / np.sqrt(np.mean(ref * ref))))
    print('M=%d N=%d K=%d ta=%d tb=%d  rms (bias): ' % (M, N, K, ta, tb) + '  '.join('%s %.2e (%+.1e)' % (n, r, bb) for n, (r, bb) in zip(('f32mfma', 'x3/8', 'x3/6', 'bf16', 'numpy-f32'), out)), flush=True)

"""Edge cases of the hot path on the GPU against the oracle: single frame / single row, a row that is
fully masked inside a window (what TBPTT windows past an utterance's end look like), ragged batch
chunks (B = 33 -> chunks of 16, 16, 1), and the C-ABI's argument / workspace errors."""
import ctypes as C

import numpy as np
import pytest
import torch

from common import make_batch
from test_gpu_parity import run_case, perturb_params, _dev

pytestmark = pytest.mark.gpu


def _nets():
    from asr_amd.builders import build_deep_lstm, build_deep_hyperlstm
    from asr_amd.gating_hyper_nets import (deep_projection_ln_lstm_model, deep_scaling_hyper_model,
                                           deep_projection_hyper_model)
    yield 'lstm', lambda: build_deep_lstm(None, None, 5, 2, 8, 4, 6, 1.0, True, False, 0, device=_dev(), seed=51)
    yield 'hyper', lambda: build_deep_hyperlstm('h', None, None, 5, 2, 8, 4, 2, 6, 1.0, True, 0, device=_dev(), seed=52)
    yield 'ln_prj', lambda: deep_projection_ln_lstm_model(None, None, 5, 6, 2, 4, 8, grad_clipping=1.0, dropout=0.0,
                                                          device=_dev(), seed=53)
    yield 'scaling', lambda: deep_scaling_hyper_model(None, None, 5, 6, [4], [8], grad_clipping=1.0, device=_dev(), seed=54)
    yield 'prjhyper', lambda: deep_projection_hyper_model(None, None, 5, 6, 2, 4, 8, grad_clipping=1.0, device=_dev(),
                                                          seed=55)[0]


@pytest.mark.parametrize('name', ['lstm', 'hyper', 'ln_prj', 'scaling', 'prjhyper'])
@pytest.mark.parametrize('B,T,lengths', [(1, 1, [1]), (2, 5, [5, 0]), (33, 3, None), (17, 2, [2] * 16 + [1])])
def test_edge_shapes(name, B, T, lengths):
    net = dict(_nets())[name]()
    perturb_params(net, 60, 0.1)
    b = make_batch(np.random.default_rng(61), B, T, 5, 6, lengths=lengths)
    if lengths is not None and 0 in lengths:
        assert b['mask'][lengths.index(0)].sum() == 0
    run_case(net, b)
    if lengths is not None and 0 in lengths:              # the masked row contributes no gradient and no loss
        b2 = {k: (v[:1] if hasattr(v, 'shape') and v.shape[0] == B else v) for k, v in b.items()}
        net.loss(*(torch.from_numpy(b[k]).to(net.device) for k in ('x', 'mask', 'y', 'mask')))
        net.backward()
        g_full = [g.copy() for g in net.get_all_grad_values()]
        net.loss(*(torch.from_numpy(b2[k]).to(net.device) for k in ('x', 'mask', 'y', 'mask')))
        net.backward()
        for a, c in zip(g_full, net.get_all_grad_values()):
            # ce_cost = sum(ce) / B: the one-row batch scales every gradient by B (=2)
            assert np.allclose(a * B, c, rtol=2e-5, atol=1e-7)


def test_c_abi_errors():
    from asr_amd import _lib
    L = _lib.lib()
    assert L.asr_gemm_f32(0, 0, 4, 4, 4, 1.0, None, 4, None, 4, 0.0, None, 4, None, None) == -1
    d = _lib.HyperScanDesc()
    d.T, d.B, d.N, d.H, d.ndir = 4, 2, 10, 4, 1          # N % 4 != 0
    ws = torch.zeros(1 << 22, dtype=torch.uint8, device=_dev())
    assert L.asr_hyper_scan_fwd(C.byref(d), ws.data_ptr(), ws.numel(), None) == -1
    d.N = 8; d.H = 12                                     # H > N
    assert L.asr_hyper_scan_fwd(C.byref(d), ws.data_ptr(), ws.numel(), None) == -1
    d.H = 4; d.lhuc_reparam = 9
    m = torch.ones(2, 4, device=_dev()); d.mask = m.data_ptr()
    assert L.asr_hyper_scan_fwd(C.byref(d), ws.data_ptr(), ws.numel(), None) == -1
    ld = _lib.LstmScanDesc()
    ld.T, ld.B, ld.N, ld.ndir = 4, 2, 8, 1
    ld.mask = m.data_ptr()
    x = torch.zeros(8, 32, device=_dev())
    for k in ('xin', 'wpack', 'r_out', 'cst'):
        setattr(ld.dir[0], k, x.data_ptr())
    assert L.asr_lstm_scan_fwd(C.byref(ld), ws.data_ptr(), 16, None) == -3
    pd = _lib.PrjHyperDesc()
    pd.T, pd.B, pd.N, pd.F, pd.ndir, pd.Btot = 4, 65, 8, 4, 1, 65     # more than 64 rows in one call
    pd.mask = m.data_ptr()
    assert L.asr_prjhyper_fwd(C.byref(pd), ws.data_ptr(), ws.numel(), None) == -1
    pd.B = 2; pd.F = 6                                                # F % 4 != 0
    assert L.asr_prjhyper_fwd(C.byref(pd), ws.data_ptr(), ws.numel(), None) == -1
    pd.F = 4
    assert L.asr_prjhyper_fwd(C.byref(pd), ws.data_ptr(), 64, None) == -3
    assert L.asr_prjhyper_fwd(C.byref(pd), ws.data_ptr(), ws.numel(), None) == -1   # null tensors
    assert L.asr_opt_step(9, ws.data_ptr(), ws.data_ptr(), None, None, 4, 0.1, 0.0, 0.0, 0.0, None, None) == -1
    assert L.asr_opt_step(1, ws.data_ptr(), ws.data_ptr(), None, None, 4, 0.1, 0.9, 0.0, 0.0, None, None) == -1

"""CPU tests of the host side: C-ABI surface, Kaldi posterior files, pickles, TBPTT windowing, flags,
batch streams, parameter inventories and the data-parallel helpers (gloo, world_size 2)."""
import ctypes as C
import io
import os
import pickle
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# ------------------------------------------------------------------------------------------------
# C-ABI: header <-> library <-> ctypes binding
def _header_functions():
    src = open(os.path.join(ROOT, 'include', 'asr_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(asr_[a-z0-9_]+)\s*\(', src)))


def test_header_declares_every_bound_symbol():
    from asr_amd import _lib
    assert _header_functions() == _lib.declared_symbols()


def test_library_loads_and_exports_every_declared_symbol():
    from asr_amd import _lib
    L = _lib.lib()                       # resolves every symbol of _SIGS; no compute call
    for name in _header_functions():
        assert hasattr(L, name), name
    # pure host helpers can be called without a GPU
    assert L.asr_lstm_pack_floats(256, 0) > 0
    assert L.asr_hyper_pack_floats(512, 128, 0) > 0
    assert L.asr_hyper_scan_ws_bytes(100, 16, 512, 128, 2, 1) > L.asr_hyper_scan_ws_bytes(100, 16, 512, 128, 2, 0)
    assert L.asr_colreduce_ws_bytes(2048, 0) > 0


def test_header_is_plain_c_and_struct_layout_matches_ctypes(tmp_path):
    from asr_amd import _lib
    prog = tmp_path / 'layout.c'
    prog.write_text('''
#include <stdio.h>
#include <stddef.h>
#include "asr_amd.h"
int main(void) {
  printf("%zu %zu %zu %zu\\n", sizeof(asr_lstm_dir), sizeof(asr_lstm_scan_desc), sizeof(asr_hyper_dir),
         sizeof(asr_hyper_scan_desc));
  printf("%zu %zu %zu %zu\\n", offsetof(asr_lstm_scan_desc, dir), offsetof(asr_lstm_dir, dOut),
         offsetof(asr_hyper_scan_desc, dir), offsetof(asr_hyper_dir, dD));
  printf("%zu %zu %zu %zu\\n", sizeof(asr_prjhyper_dir), sizeof(asr_prjhyper_desc), offsetof(asr_prjhyper_desc, dir),
         offsetof(asr_prjhyper_dir, douto));
  printf("%zu %zu %zu %zu\\n", sizeof(asr_scaling_dir), sizeof(asr_scaling_desc), offsetof(asr_scaling_desc, dir),
         offsetof(asr_scaling_dir, dinit));
  return 0;
}''')
    exe = tmp_path / 'layout'
    subprocess.check_call(['gcc', '-std=c99', '-Wall', '-Werror', '-I', os.path.join(ROOT, 'include'),
                           str(prog), '-o', str(exe)])
    out = subprocess.check_output([str(exe)]).decode().split()
    got = [int(v) for v in out]
    want = [C.sizeof(_lib.LstmDir), C.sizeof(_lib.LstmScanDesc), C.sizeof(_lib.HyperDir),
            C.sizeof(_lib.HyperScanDesc), _lib.LstmScanDesc.dir.offset, _lib.LstmDir.dOut.offset,
            _lib.HyperScanDesc.dir.offset, _lib.HyperDir.dD.offset,
            C.sizeof(_lib.PrjHyperDir), C.sizeof(_lib.PrjHyperDesc), _lib.PrjHyperDesc.dir.offset,
            _lib.PrjHyperDir.douto.offset,
            C.sizeof(_lib.ScalingDir), C.sizeof(_lib.ScalingDesc), _lib.ScalingDesc.dir.offset,
            _lib.ScalingDir.dinit.offset]
    assert got == want


def test_missing_library_fails_loudly(tmp_path):
    code = ("import os, sys; sys.path.insert(0, %r); os.environ['ASR_AMD_LIB'] = %r\n"
            "from asr_amd import _lib\n"
            "try:\n    _lib.lib()\nexcept RuntimeError as e:\n    print('LOUD', e)\n" % (ROOT, str(tmp_path / 'nope.so')))
    out = subprocess.check_output([sys.executable, '-c', code]).decode()
    assert 'LOUD' in out and 'no CPU fallback' in out


# ------------------------------------------------------------------------------------------------
# Kaldi posterior files
def test_kaldi_ark_binary_layout_and_roundtrip(tmp_path):
    from asr_amd import kaldi_io
    rng = np.random.default_rng(0)
    mats = [('utt_a', rng.standard_normal((5, 7)).astype(np.float32)),
            ('utt_b', rng.standard_normal((1, 7)).astype(np.float32)),
            ('utt_c', np.zeros((0, 7), dtype=np.float32))]
    p = str(tmp_path / 'post.ark')
    with kaldi_io.BaseFloatMatrixWriter('ark:' + p) as w:
        for k, m in mats:
            w.write(k.encode('ascii'), m)          # the reference passes uttid.encode('ascii')
    raw = open(p, 'rb').read()
    # byte-exact header of the first entry: key, space, \0B, FM, space, \4 rows, \4 cols
    assert raw.startswith(b'utt_a \x00BFM \x04\x05\x00\x00\x00\x04\x07\x00\x00\x00')
    assert raw[21:21 + 5 * 7 * 4] == mats[0][1].tobytes()
    back = kaldi_io.read_mat_ark('ark:' + p)
    assert [k for k, _ in back] == [k for k, _ in mats]
    for (_, a), (_, b) in zip(back, mats):
        assert a.shape == b.shape and np.array_equal(a, b)


def test_kaldi_ark_scp_and_text(tmp_path):
    from asr_amd import kaldi_io
    rng = np.random.default_rng(1)
    mats = [('u%d' % i, rng.standard_normal((i + 2, 3)).astype(np.float32)) for i in range(4)]
    ark, scp, txt = (str(tmp_path / n) for n in ('a.ark', 'a.scp', 'a.txt'))
    with kaldi_io.BaseFloatMatrixWriter('ark,scp:%s,%s' % (ark, scp)) as w:
        for k, m in mats:
            w.write(k, m)
    back = kaldi_io.read_mat_ark('scp:' + scp)
    assert all(np.array_equal(a, b) for (_, a), (_, b) in zip(back, mats))
    with kaldi_io.BaseFloatMatrixWriter('ark,t:' + txt) as w:
        for k, m in mats:
            w.write(k, m)
    back = kaldi_io.read_mat_ark('ark:' + txt)
    assert [k for k, _ in back] == [k for k, _ in mats]
    assert all(np.array_equal(a, b) for (_, a), (_, b) in zip(back, mats))   # repr() round-trips f32
    with pytest.raises(ValueError):
        kaldi_io.BaseFloatMatrixWriter('ark:' + ark).write('bad key', mats[0][1])
    with pytest.raises(ValueError):
        kaldi_io.BaseFloatMatrixWriter('ark:' + ark).write('k', np.zeros(3))


# ------------------------------------------------------------------------------------------------
# pickles / parameter lists
class _FakeParam(object):
    def __init__(self, v):
        self.v = np.array(v, dtype=np.float32)

    def get_value(self):
        return self.v.copy()

    def set_value(self, v):
        self.v = np.array(v, dtype=np.float32)


def test_model_pickle_layout_and_errors(tmp_path):
    from asr_amd import utils
    params = [_FakeParam(np.arange(6).reshape(2, 3)), _FakeParam(np.ones(4))]
    p = str(tmp_path / 'm_last_model.pkl')
    utils.save_network(params, {'t_prev': None}, 7, p)
    obj = pickle.load(open(p, 'rb'))
    assert isinstance(obj, list) and len(obj) == 3
    assert obj[1] is None and obj[2] == 7                     # reference quirk: updater values are None
    assert [a.shape for a in obj[0]] == [(2, 3), (4,)]
    for q in params:
        q.set_value(np.zeros_like(q.v))

    class A(object):
        reload_model = p
    upd, epoch = utils.load_or_init_model(params, A())
    assert upd is None and epoch == 7 and params[0].v[1, 2] == 5
    with pytest.raises(ValueError, match='got 1 values to set 2 parameters'):
        utils.set_model_param_value(params, [np.zeros((2, 3))])
    with pytest.raises(ValueError, match='parameter has shape'):
        utils.set_model_param_value(params, [np.zeros((2, 3)), np.zeros(5)])
    assert params[0].v[1, 2] == 5                              # nothing was set on failure

    class B(object):
        reload_model = None; no_reload = False; save_path = str(tmp_path / 'm')
    b = B()
    utils.find_reload_model(b)
    assert b.reload_model == p


def test_python2_style_pickle_loads(tmp_path):
    from asr_amd import utils
    # a protocol-2 pickle holding latin1 bytes strings, as Python 2 writes numpy arrays
    arr = np.arange(4, dtype=np.float32)
    p = str(tmp_path / 'py2.pkl')
    with open(p, 'wb') as f:
        pickle.dump([[arr], None, 3], f, protocol=2)
    vals, upd, epoch = utils.load_network_pickle(p)
    assert epoch == 3 and np.array_equal(vals[0], arr)


def test_model_pickle_refuses_code(tmp_path):
    """A model / history file is data: anything but numpy arrays and plain containers is refused
    before it can run (the loaders sit behind `python -m asr_amd.forward MODEL` and --reload-model)."""
    from asr_amd import utils

    class Boom(object):
        def __reduce__(self):
            return (os.system, ('touch %s' % (tmp_path / 'pwned'),))

    p = str(tmp_path / 'evil.pkl')
    for payload in ([[np.zeros(2, np.float32)], Boom(), 1], Boom()):
        with open(p, 'wb') as f:
            pickle.dump(payload, f, protocol=2)
        with pytest.raises(pickle.UnpicklingError, match='refuses'):
            utils.load_network_pickle(p)
        with pytest.raises(pickle.UnpicklingError):
            utils.EvalHistory.load(p)
    assert not (tmp_path / 'pwned').exists()
    # what the package itself writes stays readable: float32/float64/int arrays, numpy scalars, epoch
    vals = [np.arange(6, dtype=np.float32).reshape(2, 3), np.ones(3, np.float64), np.array(5, np.int32)]
    for proto in (2, pickle.HIGHEST_PROTOCOL):
        with open(p, 'wb') as f:
            pickle.dump([vals, None, np.int64(4)], f, protocol=proto)
        got, upd, epoch = utils.load_network_pickle(p)
        assert upd is None and int(epoch) == 4 and all(np.array_equal(a, b) for a, b in zip(got, vals))
    # codecs.encode is only the latin1 form Python 3 writes for bytes; EvalRecord only from the modules that define it
    class Enc(object):
        def __reduce__(self):
            import codecs
            return (codecs.encode, ('x', 'idna'))

    with open(p, 'wb') as f:
        pickle.dump(Enc(), f, protocol=2)
    with pytest.raises(pickle.UnpicklingError, match='refuses'):
        utils.load_network_pickle(p)
    with open(p, 'wb') as f:
        f.write(b'\x80\x02cos\nEvalRecord\nq\x00.')       # GLOBAL os.EvalRecord
    with pytest.raises(pickle.UnpicklingError, match='refuses'):
        utils.EvalHistory.load(p)
    h = utils.EvalHistory()
    assert h.add(1.0, (2.0, 0.5), (2.1, 0.6)) and not h.add(1.0, (2.0, 0.7), (2.1, 0.6))
    h.save(p)
    back = utils.EvalHistory.load(p)
    assert [tuple(r) for r in back] == [tuple(r) for r in h] and back.best_valid_fer() == 0.5


def test_lr_schedule_of_the_hypernet_scripts():
    from asr_amd.train import lr_schedule
    # scripts/hypernets/exp_prj_ln_lstm_fix.py:316 with Python-2 integer division
    assert [round(lr_schedule(1.0, e), 6) for e in (1, 2, 3, 4, 5, 6)] == [
        1.0, 1.0, 1.0, round(1 / 1.04, 6), round(1 / 1.04, 6), round(1 / 1.04 ** 2, 6)]


def test_gen_win_matches_reference_windowing():
    from asr_amd import utils
    B, T, D = 2, 23, 3
    x = np.arange(B * T * D, dtype=np.float32).reshape(B, T, D)
    m = np.ones((B, T), dtype=np.float32)
    y = np.arange(B * T, dtype=np.int32).reshape(B, T)
    wins = list(utils.gen_win((x, m, x, m, y, m), 10, right_context=4))
    assert [w[0].shape[1] for w in wins] == [14, 13, 3]        # 0:14, 10:24 (clipped to 23), 20:23
    assert np.array_equal(wins[1][0], x[:, 10:24]) and np.array_equal(wins[2][4], y[:, 20:23])
    wins = list(utils.gen_win_test((x, m, None, None), 8))
    assert [w[0].shape[1] for w in wins] == [8, 8, 7] and wins[0][2] is None


def test_arg_parsers_and_save_path():
    from asr_amd import arg_utils
    a = arg_utils.get_arg_parser('deep_lstm').parse_args(
        ['--num-units', '256', '--num-layers', '3', '--uni', '--num-tbptt-steps', '100', '--right-context', '5',
         '--data-path', '/x/wsj_fbank123.h5', '--batch-size', '16'])
    assert a.num_units == 256 and a.uni and not a.use_layer_norm and a.grad_clipping == 1.0
    assert arg_utils.get_save_path(a) == 'wsj_fbank123_lr0.001_gc1.0_l3_n256_b16_uni_tb100_rc5_train_si284'
    h = arg_utils.get_arg_parser('hyperlstm').parse_args(['--num-nodes', '512', '--num-hyper-nodes', '128',
                                                         '--num-proj-nodes', '64', '--num-layers', '4'])
    assert arg_utils.get_save_path(h, 'hyperlstm') == './wsj_hyperlstm_lr0.001_gc1.0_l4_n512_b2_hn128_pn64'
    l = arg_utils.get_arg_parser('lhuclstm').parse_args([])
    assert l.num_pred_units == 128 and l.num_pred_layers == 1 and l.layer_name == 'IVectorLHUCLSTMLayer'


def test_utterance_stream_pads_like_fuel():
    from asr_amd import data
    utts = data.synthetic_set(5, D=4, C=7, I=3, min_len=2, max_len=9, seed=3)
    ds = data.UtteranceStream(utts, 2)
    batches = list(ds.get_epoch_iterator())
    assert len(batches) == 3 == len(ds)
    x, xm, iv, ivm, y, ym = batches[0]
    lens = [u.features.shape[0] for u in utts[:2]]
    assert x.shape == (2, max(lens), 4) and iv.shape == (2, max(lens), 3) and y.dtype == np.int32
    assert np.array_equal(xm.sum(axis=1), lens) and np.array_equal(xm, ym)
    for i, u in enumerate(utts[:2]):
        assert np.array_equal(x[i, :lens[i]], u.features) and not x[i, lens[i]:].any()
        assert np.array_equal(iv[i, 0], iv[i, lens[i] - 1])    # constant within the utterance
    ids = list(data.UttIdStream(utts, 2).get_epoch_iterator())
    assert ids[0] == (['utt0000', 'utt0001'],) and len(ids[2][0]) == 1
    assert len(list(data.UtteranceStream(utts, 2, test=True).get_epoch_iterator())[0]) == 4


# ------------------------------------------------------------------------------------------------
# parameter inventories (pickle order) -- pure python
def test_param_inventories_match_survey_counts():
    from asr_amd import specs
    def count(sp):
        return sum(int(np.prod(s)) for _, s, _ in sp)
    # cfg3 per direction L1 2,185,728, L2-4 4,492,288; output 3,521,900; total 34.85 M
    assert count(specs.hyper_params(123, 512, 128, 64)) == 2185728
    assert count(specs.hyper_params(1024, 512, 128, 64)) == 4492288
    # ScalingHyperLSTMLayer, trainable set at use_peepholes=False / use_layer_norm=False
    D_, H_, N_ = 123, 250, 500
    assert count(specs.scaling_hyper_params(D_, H_, N_)) == 4 * (D_ * H_ + H_ * H_ + N_ * H_ + H_) + 4 * (D_ * N_ + N_ * N_ + 3 * (H_ * N_ + N_))
    assert len(specs.scaling_hyper_params(5, 6, 8)) == 48
    assert count(specs.dense_params(1024, 3436)) == 3521900
    names = [n for n, _, _ in specs.hyper_params(5, 8, 4, 2)]
    assert names[:3] == ['W_hhat_ig', 'W_xhat_ig', 'bhat_ig'] and names[12:17] == [
        'W_hhat_h_ig', 'b_hhat_h_ig', 'W_hhat_x_ig', 'b_hhat_x_ig', 'W_hhat_b_ig']
    assert names[-2:] == ['W_h_og', 'W_x_og'] and len(names) == 12 + 20 + 16 + 8
    names = [n for n, _, _ in specs.lstm_params(5, 8, 4, True)]
    assert names[:3] == ['W_h_ig', 'W_x_ig', 'b_ig'] and names[12:] == [
        'W_p', 'W_x_alpha', 'W_h_alpha', 'W_c_alpha', 'W_c_beta']
    shapes = dict((n, s) for n, s, _ in specs.lstm_params(5, 8, 4, True))
    assert shapes['W_h_ig'] == (4, 8) and shapes['W_p'] == (8, 4)
    names = [n for n, _, _ in specs.icond_ln_prj_params(5, 8, 3, 6)]
    assert names == ['W_in_to_ingate', 'W_hid_to_ingate', 'W_in_to_forgetgate', 'W_hid_to_forgetgate',
                     'W_in_to_cell', 'W_hid_to_cell', 'W_in_to_outgate', 'W_hid_to_outgate', 'W_hid_prj',
                     'W_cond_prj', 'b_cond_prj', 'W_alpha_in', 'b_alpha_in', 'W_alpha_hid', 'b_alpha_hid',
                     'W_beta', 'b_beta']
    names = [n for n, _, _ in specs.ln_prj_params(5, 8, 4)]
    assert names[-7:] == ['W_alpha_in', 'W_beta_in', 'W_alpha_hid', 'W_beta_hid', 'W_alpha_cell',
                          'W_beta_cell', 'W_hid_prj']


# ------------------------------------------------------------------------------------------------
# data parallel: sharding + bucketed all-reduce over gloo, world_size 2
def test_shard_rows_partition():
    from asr_amd import dp
    lens = np.array([50, 10, 40, 20, 30, 60, 70, 5])
    got = [dp.shard_rows(lens, r, 4) for r in range(4)]
    assert sorted(np.concatenate(got).tolist()) == list(range(8))
    assert [lens[g].tolist() for g in got] == [[5, 10], [20, 30], [40, 50], [60, 70]]
    with pytest.raises(ValueError):
        dp.shard_rows(lens, 0, 3)
    x = np.zeros((8, 70, 2), dtype=np.float32)
    m = (np.arange(70)[None, :] < lens[:, None]).astype(np.float32)
    sx, sm = dp.shard_batch((x, m), 1, 4)
    assert sx.shape == (2, 30, 2) and sm.sum() == 50


_GLOO_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import numpy as np, torch, torch.distributed as dist
from asr_amd import dp

class Store(object):
    pass

class FakeNet(object):
    """The slice of engine.Network that GradAllReduce touches, on the CPU."""
    def __init__(self, n):
        self.device = torch.device('cpu')
        self.store = Store(); self.store.g = torch.zeros(n)
        self.world = 1; self.grad_hook = None

rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
net = FakeNet(1000)
ar = dp.GradAllReduce(net)
assert net.world == 2 and net.grad_hook is not None
rng = np.random.default_rng(7)
full = [torch.from_numpy(rng.standard_normal(1000).astype(np.float32)) for _ in range(world)]
net.store.g.copy_(full[rank])
# buckets become final in reverse topological order (output layer first)
for lo, hi in ((900, 1000), (500, 900), (0, 500)):
    net.grad_hook(lo, hi)
ar.finish()
want = full[0] + full[1]
assert torch.equal(net.store.g, want), (net.store.g - want).abs().max()
s = ar.reduce_scalars(torch.tensor([1.0 + rank, 10.0]))
assert s.tolist() == [3.0, 20.0]
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'ok')
'''


def test_gloo_world2_bucketed_allreduce(tmp_path):
    script = tmp_path / 'worker.py'
    script.write_text(_GLOO_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29531', WORLD_SIZE='2')
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        assert 'rank %d ok' % r in o


_GLOO_STATUS_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r)
import torch, torch.distributed as dist
from asr_amd import dp
from asr_amd._lib import AsrError

class Store(object):
    pass

class FakeNet(object):
    """engine.Network as GradAllReduce sees it; check_status() fails on the rank named by FAIL_RANK."""
    def __init__(self, rank):
        self.device = torch.device('cpu'); self.rank = rank
        self.store = Store(); self.store.g = torch.full((64,), float(rank + 1)); self.store.p = torch.zeros(64)
        self.world = 1; self.grad_hook = None
    def check_status(self):
        if os.environ.get('FAIL_RANK') == str(self.rank):
            raise AsrError('persistent scan gave up: ASR_ERR_TIMEOUT (-4)')

rank = int(os.environ['RANK']); world = int(os.environ['WORLD_SIZE'])
dist.init_process_group('gloo', rank=rank, world_size=world)
net = FakeNet(rank)
ar = dp.GradAllReduce(net)
assert ar.overlap == (os.environ.get('ASR_DP_OVERLAP', '1') != '0')
assert (net.grad_hook is not None) == ar.overlap
if net.grad_hook:
    net.grad_hook(0, 64)
ar.finish()
assert net.store.g.tolist() == [3.0] * 64            # summed with or without overlap
stepped = False
try:
    ar.agree_on_status()
    net.store.p -= 0.1 * net.store.g                 # the optimiser step: only if EVERY rank's scans were fine
    stepped = True
except AsrError as e:
    print('rank', rank, 'raised:', e)
dist.barrier(); dist.destroy_process_group()
print('rank', rank, 'stepped' if stepped else 'skipped')
'''


@pytest.mark.parametrize('fail_rank,overlap', [('1', '1'), ('0', '0'), ('', '0')])
def test_gloo_world2_status_agreement_and_no_overlap_fallback(tmp_path, fail_rank, overlap):
    """dp.GradAllReduce.agree_on_status: when ONE rank's persistent scan reports a timeout, BOTH ranks raise before
    the optimiser step (no rank steps on garbage, none is left inside a collective); with ASR_DP_OVERLAP=0 the whole
    gradient is reduced in finish() and the sums are the same."""
    script = tmp_path / 'worker.py'
    script.write_text(_GLOO_STATUS_WORKER % dict(root=ROOT))
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT='29537', WORLD_SIZE='2', FAIL_RANK=fail_rank,
               ASR_DP_OVERLAP=overlap)
    procs = [subprocess.Popen([sys.executable, str(script)], env=dict(env, RANK=str(r)),
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT) for r in range(2)]
    outs = [p.communicate(timeout=120)[0].decode() for p in procs]
    for r, (p, o) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, o
        if fail_rank:
            assert 'rank %d raised' % r in o and 'rank %d skipped' % r in o, o
            if str(r) != fail_rank:
                assert 'another rank' in o, o
        else:
            assert 'rank %d stepped' % r in o, o


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` outside torch.distributed.run spawns two ranks itself (before any GPU
    call) and relays rank 0's single JSON line; rehearsed on the CPU over gloo (--selftest-cpu)."""
    import json
    env = dict((k, v) for k, v in os.environ.items() if k not in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'))
    out = subprocess.check_output([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--selftest-cpu',
                                   '--steps', '3', '--warmup', '1'], env=env, timeout=300).decode()
    lines = [l for l in out.splitlines() if l.startswith('{')]
    assert len(lines) == 1, out
    rec = json.loads(lines[0])
    assert rec['n_gpus'] == 2 and rec['config']['parallelism'] == 'dp2' and rec['config']['global_batch'] == 32
    assert rec['steps'] == 3 and rec['warmup'] == 1 and rec['selftest'] is True
    assert 'ms_per_step_median' in rec and rec['scaling'] == 'weak'
    # a rank count that disagrees with the launcher's world size is refused, not silently measured
    env2 = dict(env, WORLD_SIZE='1', RANK='0', LOCAL_RANK='0')
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--selftest-cpu'], env=env2,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b'WORLD_SIZE=1' in r.stderr
    # a rank that dies (here before the rendezvous) ends the run: the launcher stops the ranks still waiting for it
    # and exits non-zero instead of sitting in the collective's own timeout
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--selftest-cpu', '--steps', '2',
                        '--warmup', '1'], env=dict(env, ASR_BENCH_SELFTEST_FAIL_RANK='1'), stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=120)
    assert r.returncode != 0 and b'ranks failed' in r.stderr and time.time() - t0 < 60


def test_oracle_updaters_known_answers():
    """Hand-computed steps of libs/lasagne_libs/updates.py formulas (momentum :143-147, nesterov
    :244-249, adamax :662-696, total_norm_constraint :848-855)."""
    from oracle import asr_oracle as O
    p = [np.array([1.0, -2.0])]; g = [np.array([0.5, 0.25])]; v = [np.zeros(2)]
    O.opt_step('momentum', p, g, v, None, 0.1, 0.9)
    assert np.allclose(v[0], [-0.05, -0.025]) and np.allclose(p[0], [0.95, -2.025])
    O.opt_step('momentum', p, g, v, None, 0.1, 0.9)
    assert np.allclose(v[0], [-0.095, -0.0475]) and np.allclose(p[0], [0.855, -2.0725])
    p = [np.array([1.0])]; v = [np.zeros(1)]
    O.opt_step('nesterov', p, [np.array([1.0])], v, None, 0.1, 0.9)
    assert np.allclose(v[0], [-0.1]) and np.allclose(p[0], [1.0 - 0.09 - 0.1])
    p = [np.array([1.0])]; m = [np.zeros(1)]; u = [np.zeros(1)]
    t = O.opt_step('adamax', p, [np.array([2.0])], m, u, 0.002, 0.9, 0.999, 1e-8, t_prev=0)
    assert t == 1 and np.allclose(m[0], [0.2]) and np.allclose(u[0], [2.0])
    assert np.allclose(p[0], [1.0 - (0.002 / 0.1) * 0.2 / (2.0 + 1e-8)])
    sg, norm = O.total_norm_constraint([np.array([3.0]), np.array([4.0])], 1.0)
    assert abs(norm - 5.0) < 1e-12 and np.allclose(np.concatenate(sg), [0.6, 0.8], atol=1e-6)
    sg, norm = O.total_norm_constraint([np.array([0.3]), np.array([0.4])], 1.0)
    assert np.allclose(np.concatenate(sg), [0.3, 0.4], atol=1e-6)


def test_updates_api_shapes():
    from asr_amd import updates as U
    for f in (U.sgd, U.momentum, U.nesterov_momentum, U.adagrad, U.rmsprop, U.adadelta, U.adamax, U.adam):
        upd, d = f(loss_or_grads=None, params=[], learning_rate=0.1)
        assert hasattr(upd, 'apply')
    upd, d = U.adamax(None, [], 0.002, load_params_dict={'t_prev': np.float32(7)})
    assert float(d['t_prev'].get_value()) == 7.0
    cg, nh = U.total_norm_constraint(None, max_norm=5.0, return_norm=True)
    upd, d = U.adam(loss_or_grads=cg, params=[], learning_rate=0.1)
    assert upd.kind == 7 and upd.clip.max_norm == 5.0


def test_transformers():
    """data/transformers.py:15-63,133-177; libs/utils.py:208-262; bin/forward_lstm2.py:99-101."""
    from asr_amd import transformers as TR

    class S(object):
        def __init__(self, batches): self.b = batches
        def get_epoch_iterator(self): return iter(self.b)

    f = [np.arange(6, dtype=np.float32).reshape(3, 2), np.arange(4, dtype=np.float32).reshape(2, 2)]
    iv = [np.ones((3, 1), np.float32), 2 * np.ones((2, 1), np.float32)]
    tg = [np.array([1, 2, 3], np.int32), np.array([4, 5], np.int32)]
    out = next(iter(TR.DelayTransformer(S([(f, iv, tg)]), 2).get_epoch_iterator()))
    assert out[0][0].shape == (5, 2) and (out[0][0][3:] == f[0][-1]).all()
    assert out[1][1].shape == (4, 1) and (out[1][1] == 2).all()
    assert out[2][0].tolist() == [0, 0, 1, 2, 3] and out[2][1].tolist() == [0, 0, 4, 5]
    # length sort: batches of 1 example each, cache 4 -> sorted runs of 4, then the rest
    lens = [5, 2, 9, 3, 7, 1]
    st = S([([np.zeros((n, 1))],) for n in lens])
    got = [len(b[0][0]) for b in TR.LengthSortTransformer(st, 2, 4).get_epoch_iterator() for _ in [0]]
    allg = [len(x) for b in TR.LengthSortTransformer(st, 2, 4).get_epoch_iterator() for x in b[0]]
    assert sorted(allg) == sorted(lens) and allg[:4] == [2, 3, 5, 9]
    with pytest.raises(ValueError):
        TR.LengthSortTransformer(st, 4, 2)
    sk = TR.skip_frames([[np.arange(7)], [np.arange(14).reshape(7, 2)]], 3)
    assert sk[0][0].tolist() == [0, 3, 6] and sk[1][0].shape == (3, 2)
    fx = list(TR.skip_frames_fixed([[np.arange(7)]], 2, return_first=True))
    assert fx[0][0][0].tolist() == [0, 2, 4, 6]
    assert list(TR.skip_frames_fixed([[np.arange(3)]], 1))[0][0][0].tolist() == [0, 1, 2]
    o = TR.unskip(np.arange(6, dtype=np.float32).reshape(1, 3, 2), 3, 7)
    assert o.shape == (1, 7, 2) and (o[0, 3] == o[0, 5]).all() and (o[0, 6] == [4, 5]).all()


def test_reference_held_skip_frame_vectors():
    """The only fixtures the reference's tests hold for anything built here (test/test_libs_utils.py:22-51),
    kept as data in tests/golden/ref_test_libs_utils.json.  testSkipFamesFixed lists one answer per start
    offset 0, 1, 2, but the reference's current skip_frames_fixed (libs/utils.py:231-259) yields ONE
    sub-sequence (offset 0 with return_first, else a random offset); we follow the current function and
    reproduce each listed answer by pinning the offset."""
    import json
    import random
    from asr_amd import transformers as TR
    g = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'ref_test_libs_utils.json')))
    a_seq = np.array(g['a_seq']); n = g['every_n']
    got = TR.skip_frames(a_seq, n, False)
    assert np.array_equal(got[0][0], np.array(g['skip_frames']))
    fx = list(TR.skip_frames_fixed(a_seq, n, return_first=True))
    assert len(fx) == 1 and np.array_equal(fx[0][0][0], np.array(g['skip_frames_fixed_by_start'][0]))
    real = random.randint
    try:
        for start, want in enumerate(g['skip_frames_fixed_by_start']):
            random.randint = lambda a, b, s=start: s
            fx = list(TR.skip_frames_fixed(a_seq, n))
            assert len(fx) == 1 and np.array_equal(fx[0][0][0], np.array(want)), start
    finally:
        random.randint = real
    # testRepeat: the un-skipping of posteriors, bin/forward_lstm2.py:99-101
    sk = TR.skip_frames(a_seq, n, False)[0][0]
    orig_len = len(a_seq[0][0])
    assert np.array_equal(np.repeat(sk, n, axis=0)[:orig_len], np.array(g['repeat_then_trim']))
    assert np.array_equal(TR.unskip(sk[None], n, orig_len)[0], np.array(g['repeat_then_trim']))
    # the random form draws one index per block of every_n frames, clipped to the last frame
    random.seed(3)
    r = TR.skip_frames(a_seq, n, True)[0][0]
    assert r.shape == (2, 2) and all(any((row == f).all() for f in a_seq[0][0][i * n:(i + 1) * n]) for i, row in enumerate(r))


def test_regularizable_ranges_cover_the_rec_family():
    """Every gating_hyper_nets builder maps lasagne's `regularizable` tag onto whole engine tensors."""
    from asr_amd import gating_hyper_nets as G
    from asr_amd.script_graphs import lasagne_regularizable
    kw = dict(device='cpu', seed=1)
    nets = [G.deep_projection_ln_lstm_model(None, None, 5, 6, 2, 4, 8, 1, 0.0, **kw),
            G.deep_projection_lstm_model(None, None, 5, 6, 2, 4, 8, 1, 0.0, **kw),
            G.deep_projection_ivector_ln_model(None, None, None, 5, 6, 2, 1, 3, 8, 1, 0.0, cond_dim=3, **kw)[0],
            G.deep_projection_cond_ln_model(None, None, 5, 6, 2, 1, 3, 8, 1, 0.0, **kw)[0],
            G.deep_projection_max_cond_ln_model(None, None, 5, 6, 2, 1, 3, 8, 1, 0.0, **kw)[0],
            G.deep_lstm_model(None, None, 5, 6, 2, 8, 1, 0.0, **kw),
            G.deep_ln_lstm_model(None, None, 5, 6, 2, 8, 1, 0.0, **kw),
            G.deep_prj_lstm_dln_model_v2(None, None, 5, 6, 2, 1, 3, 8, 1, 0.0, **kw)[0]
            if hasattr(G, 'deep_prj_lstm_dln_model_v2') else None]
    for net in nets:
        if net is None:
            continue
        rr = net.regularizable_ranges(lasagne_regularizable)
        assert rr and all(lo % 4 == 0 and hi > lo for lo, hi in rr)
        tot = sum(hi - lo for lo, hi in rr)
        assert 0 < tot < net.store.size


def test_eval_net_skip_scores_all_frames_with_repeated_predictions():
    """libs/comp_graph_utils.py:200-236: CE over the frames the network saw, FER over every frame with the
    predictions repeated `skip` times; checked against a direct count on a hand-made stream."""
    from asr_amd import comp_graph_utils as CG
    rng = np.random.default_rng(3)
    B, T, D, C, skip = 3, 11, 2, 4, 3
    lens = np.array([11, 4, 7])
    x = rng.standard_normal((B, T, D)).astype(np.float32)
    m = (np.arange(T)[None] < lens[:, None]).astype(np.float32)
    y = rng.integers(0, C, (B, T)).astype(np.int32)

    class Stream(object):
        def get_epoch_iterator(self):
            yield (x, m, x, m, y, m)

    seen = {}

    def predict_fn(sx, sm, sy, sym):
        seen['shape'] = sx.shape
        assert np.array_equal(sx, x[:, ::skip]) and np.array_equal(sy, y[:, ::skip])
        pred = np.where(np.arange(sx.shape[1])[None] % 2 == 0, sy, (sy + 1) % C)     # right on even skipped frames
        return 2.5 * sym.sum(), pred

    ce, fer = CG.eval_net_skip(predict_fn, Stream(), skip, False)
    assert seen['shape'] == (B, 4, D)
    assert abs(ce - 2.5) < 1e-12
    pred_full = np.repeat(np.where(np.arange(4)[None] % 2 == 0, y[:, ::skip], (y[:, ::skip] + 1) % C), skip, axis=1)[:, :T]
    want = 1.0 - ((pred_full == y) * m).sum() / m.sum()
    assert abs(fer - want) < 1e-12


def test_length_sorted_batches_are_slices_of_a_sorted_cache():
    """workloads.length_sorted_batches (the reference's LengthSortTransformer, data/transformers.py:15-63): a cache of >= 1024
    utterances is sorted by length and cut into consecutive batches; every utterance of a cache lands in exactly one batch."""
    from asr_amd import workloads
    B = 16
    bs = workloads.length_sorted_batches(3, B, 64)
    assert len(bs) == 64 and all(len(b) == B for b in bs)
    cache = np.sort(np.concatenate(bs))                      # one cache = 1024 utterances = 64 batches of 16
    again = workloads.length_sorted_batches(3, B, 64)
    assert all((a == b).all() for a, b in zip(bs, again))    # seeded
    for b in bs:
        assert (np.diff(b) >= 0).all()                       # sorted inside a batch
        i = np.searchsorted(cache, b[0], side='left')        # and a run of neighbours in the sorted cache
        spread = b[-1] - b[0]
        assert spread <= cache[min(i + 2 * B, len(cache) - 1)] - cache[i]
    spreads = [int(b[-1] - b[0]) for b in bs]
    rng = np.random.default_rng(0)
    unsorted = [workloads.lengths_for(3, rng, B) for _ in range(64)]
    assert np.mean(spreads) < 0.1 * np.mean([int(u.max() - u.min()) for u in unsorted])
    # padded frames: sorted batches waste far less than random ones
    waste = lambda L: 1.0 - sum(int(l.sum()) for l in L) / sum(int(l.max()) * len(l) for l in L)
    assert waste(bs) < 0.05 < waste(unsorted)

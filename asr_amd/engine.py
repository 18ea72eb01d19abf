"""Execution engine: runs a stacked (bi)directional recurrent acoustic model on one MI355X through
the C-ABI kernels.  torch supplies device memory and streams; all arithmetic is HIP.

Internal conventions
  * activations are time-major 2-D matrices (T*B, width), row r = t*B + b;
  * every 4N-wide gate tensor / weight is gate-interleaved: col = unit*4 + gate;
  * all trainable parameters live in ONE flat fp32 buffer (`ParamStore.p`) with a same-shaped flat
    gradient buffer, so Adam, the gradient norm and the data-parallel all-reduce are single passes;
    `ref_views` give strided views with the reference's per-gate names/shapes for pickles.
"""
import contextlib
import ctypes as C
import os
import numpy as np
import torch

from . import _lib, ops, specs
from ._lib import check

G4L = specs.GATES_LHUC
G4R = specs.GATES_REC


class Arena(object):
    """Named, growable device buffers (no per-step allocation once sizes have been seen)."""
    def __init__(self, device):
        self.device = device
        self.bufs = {}

    def get(self, name, numel, dtype=torch.float32, zero=False):
        t = self.bufs.get(name)
        if t is None or t.numel() < numel or t.dtype != dtype:
            t = torch.empty(max(int(numel), 1), dtype=dtype, device=self.device)
            self.bufs[name] = t
        v = t[:numel]
        if zero:
            v.zero_()
        return v

    def mat(self, name, rows, cols, zero=False):
        return self.get(name, rows * cols, zero=zero).view(rows, cols)


class ParamStore(object):
    def __init__(self):
        self.entries = []      # (key, shape, offset)
        self.size = 0
        self.p = None
        self.g = None

    def add(self, key, shape):
        n = int(np.prod(shape))
        self.entries.append((key, tuple(shape), self.size))
        self.size += (n + 3) // 4 * 4
        return len(self.entries) - 1

    def materialize(self, device):
        self.p = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.g = torch.zeros(self.size, dtype=torch.float32, device=device)
        self.pv = {}
        self.gv = {}
        for key, shape, off in self.entries:
            n = int(np.prod(shape))
            self.pv[key] = self.p[off:off + n].view(shape)
            self.gv[key] = self.g[off:off + n].view(shape)


class _PermView(object):
    """reference tensor (..., 4N) gate-major <-> engine (..., N, 4): copy helpers (not a plain view)."""
    def __init__(self, E):
        self.E = E

    def copy_from(self, ref):
        lead = self.E.shape[:-1]
        self.E.view(*lead, -1, 4).copy_(ref.view(*lead, 4, -1).transpose(-1, -2))

    def to_ref(self):
        lead = self.E.shape[:-1]
        return self.E.view(*lead, -1, 4).transpose(-1, -2).reshape(*lead, -1)


class DirParams(object):
    """Engine tensors of one layer-direction + the mapping to the reference's parameter list."""
    def __init__(self, store, prefix):
        self.store = store
        self.prefix = prefix
        self.keys = {}
        self.refmap = []       # (ref_name, engine_key, kind, arg)

    def add(self, name, shape):
        key = self.prefix + name
        self.store.add(key, shape)
        self.keys[name] = key

    def p(self, name):
        return self.store.pv[self.keys[name]]

    def g(self, name):
        return self.store.gv[self.keys[name]]

    def has(self, name):
        return name in self.keys

    def map(self, ref_name, name, kind='same', arg=None):
        self.refmap.append((ref_name, name, kind, arg))

    def _view(self, E, kind, arg):
        if kind == 'same':
            return E
        if kind == 'gatecols':
            return E.view(E.shape[0], -1, 4)[:, :, arg]
        if kind == 'gatevec':
            return E.view(-1, 4)[:, arg]
        if kind == 'index':
            return E[arg]
        if kind == 'aug_w':                      # (4, H+1, Z): rows < H of gate arg
            return E[arg, :-1, :]
        if kind == 'aug_b':
            return E[arg, -1, :]
        if kind == 'fn':                         # arg: callable engine tensor -> strided view
            return arg(E)
        raise KeyError(kind)

    def load_ref(self, values, grads=False):
        """values: dict ref_name -> torch tensor (device)."""
        for ref_name, name, kind, arg in self.refmap:
            E = self.g(name) if grads else self.p(name)
            v = values[ref_name]
            if kind == 'perm4':
                _PermView(E).copy_from(v)
            else:
                self._view(E, kind, arg).copy_(v)

    def ref_view(self, i, grads=False):
        """The i-th reference parameter of this direction as a numpy array (a copy)."""
        ref_name, name, kind, arg = self.refmap[i]
        E = self.g(name) if grads else self.p(name)
        v = _PermView(E).to_ref() if kind == 'perm4' else self._view(E, kind, arg)
        return v.detach().contiguous().cpu().numpy()

    def ref_set(self, i, value):
        ref_name, name, kind, arg = self.refmap[i]
        E = self.p(name)
        v = torch.from_numpy(np.ascontiguousarray(value, dtype=np.float32)).to(E.device)
        if kind == 'perm4':
            _PermView(E).copy_from(v)
        else:
            self._view(E, kind, arg).copy_(v)

    def dump_ref(self, grads=False):
        out = []
        for ref_name, name, kind, arg in self.refmap:
            E = self.g(name) if grads else self.p(name)
            v = _PermView(E).to_ref() if kind == 'perm4' else self._view(E, kind, arg)
            out.append((ref_name, v.detach().contiguous().cpu().numpy()))
        return out


class ParamHandle(object):
    """Stand-in for a Theano shared variable of the reference's parameter list."""
    def __init__(self, dp, i):
        self.dp = dp; self.i = i
        self.name = dp.prefix + dp.refmap[i][0]

    def get_value(self, borrow=False):
        return self.dp.ref_view(self.i)

    def set_value(self, value, borrow=False):
        if tuple(np.shape(value)) != tuple(self.get_value().shape):
            raise ValueError('shape mismatch for %s' % self.name)
        self.dp.ref_set(self.i, value)

    def __repr__(self):
        return self.name


def _mk_lstm_family(store, prefix, kind, D, N, P, use_ln, I=0, F=0, npu=0, npl=0):
    """Registers engine tensors and the reference mapping for one direction.  P = output width of the
    recurrence fed back through W_h (0: no projection)."""
    dp = DirParams(store, prefix)
    K = P if P else N
    dp.add('Wx', (D, 4 * N)); dp.add('Wh', (K, 4 * N))
    if kind in ('lstm', 'lhuc'):                 # libs/lhuclstm_layers.py:32-71
        dp.add('b', (4 * N,))
        for g, gn in enumerate(G4L):
            dp.map('W_h_' + gn, 'Wh', 'gatecols', g); dp.map('W_x_' + gn, 'Wx', 'gatecols', g)
            dp.map('b_' + gn, 'b', 'gatevec', g)
        if P:
            dp.add('Wp', (N, P)); dp.map('W_p', 'Wp')
        if use_ln:
            dp.add('a_in', (4 * N,)); dp.add('a_hid', (4 * N,)); dp.add('a_c', (N,)); dp.add('b_c', (N,))
            dp.map('W_x_alpha', 'a_in', 'perm4'); dp.map('W_h_alpha', 'a_hid', 'perm4')
            dp.map('W_c_alpha', 'a_c'); dp.map('W_c_beta', 'b_c')
        if kind == 'lhuc':                       # libs/lhuclstm_layers.py:327-348
            for i in range(npl + 1):
                ind = I if i == 0 else npu
                outd = npu if i < npl else (P if P else N)
                dp.add('We%d' % i, (ind, outd)); dp.add('be%d' % i, (outd,))
                dp.map('W_e_h_%d' % i, 'We%d' % i); dp.map('b_e_h_%d' % i, 'be%d' % i)
    elif kind in ('prj', 'ln_prj'):              # recurrent_layers.py:329-377, :585-647
        dp.add('b', (4 * N,))
        for g, gn in enumerate(G4R):
            dp.map('W_in_to_' + gn, 'Wx', 'gatecols', g); dp.map('W_hid_to_' + gn, 'Wh', 'gatecols', g)
            dp.map('b_' + gn, 'b', 'gatevec', g)
        if kind == 'ln_prj':
            for nm in ('a_in', 'b_in', 'a_hid', 'b_hid'):
                dp.add(nm, (4 * N,))
            dp.add('a_c', (N,)); dp.add('b_c', (N,))
            dp.map('W_alpha_in', 'a_in', 'perm4'); dp.map('W_beta_in', 'b_in', 'perm4')
            dp.map('W_alpha_hid', 'a_hid', 'perm4'); dp.map('W_beta_hid', 'b_hid', 'perm4')
            dp.map('W_alpha_cell', 'a_c'); dp.map('W_beta_cell', 'b_c')
        dp.add('Wp', (N, P)); dp.map('W_hid_prj', 'Wp')
    elif kind == 'icond_ln_prj':                 # recurrent_layers.py:1115-1188
        for g, gn in enumerate(G4R):
            dp.map('W_in_to_' + gn, 'Wx', 'gatecols', g); dp.map('W_hid_to_' + gn, 'Wh', 'gatecols', g)
        dp.add('Wp', (N, P)); dp.map('W_hid_prj', 'Wp')
        dp.add('Wc', (I, F)); dp.add('bc', (F,)); dp.map('W_cond_prj', 'Wc'); dp.map('b_cond_prj', 'bc')
        for nm, rn in (('ai', 'alpha_in'), ('ah', 'alpha_hid'), ('bt', 'beta')):
            dp.add('W' + nm, (F, 4 * N)); dp.add('b' + nm, (4 * N,))
            dp.map('W_' + rn, 'W' + nm, 'perm4'); dp.map('b_' + rn, 'b' + nm, 'perm4')
    else:
        raise KeyError(kind)
    return dp


class LstmPair(object):
    """One layer of the stack: a forward direction and optionally a backward direction of the same
    LSTM-family kind, run by ONE persistent scan launch."""
    def __init__(self, net, idx, kind, D, N, P, use_ln, clip, ndir, I=0, F=0, npu=0, npl=0,
                 stateful=False, context=0):
        self.net = net; self.idx = idx; self.kind = kind
        self.D = D; self.N = N; self.P = P; self.use_ln = use_ln; self.clip = float(clip)
        self.ndir = ndir; self.I = I; self.F = F; self.npu = npu; self.npl = npl
        self.stateful = stateful; self.context = context
        self.out_dim = ndir * (P if P else N)
        self.ln_kind = 2 if kind == 'icond_ln_prj' else (1 if (use_ln or kind == 'ln_prj') else 0)
        self.cell_ln = 1 if (kind == 'ln_prj' or (kind in ('lstm', 'lhuc') and use_ln)) else 0
        self.dirs = [_mk_lstm_family(net.store, 'L%d.%d.' % (idx, d), kind, D, N, P,
                                     self.ln_kind == 1, I, F, npu, npl) for d in range(ndir)]
        self.reverse = [False, True][:ndir]
        self.state = [None] * ndir           # TBPTT (c, r, h) per direction; h = the PROJECTED hidden state as the
                                             # window that produced it projected it (== r without a projection)

    def describe(self):
        out = {}
        for d, nm in enumerate(('fwd', 'bwd')):
            if d >= self.ndir:
                out[nm] = None
                continue
            k = self.kind
            if k in ('lstm', 'lhuc'):
                args = dict(num_units=self.N, num_proj_units=self.P, use_layer_norm=self.use_ln,
                            grad_clipping=self.clip, backwards=self.reverse[d])
                if k == 'lstm':
                    args.update(stateful=self.stateful and d == 0, context=self.context)
                else:
                    args.update(num_pred_units=self.npu, num_pred_layers=self.npl,
                                stateful=self.stateful and d == 0, context=self.context)
            elif k in ('prj', 'ln_prj'):
                args = dict(num_units=self.N, num_factors=self.P, grad_clipping=self.clip,
                            backwards=self.reverse[d])
            else:
                args = dict(num_units=self.N, num_factors=self.F, grad_clipping=self.clip,
                            backwards=self.reverse[d])
            out[nm] = dict(kind=k, args=args, params=[r[0] for r in self.dirs[d].refmap])
        return out

    # -- TBPTT state (libs/lhuclstm_layers.py:221-228)
    def reset(self, batch_size):
        """Zero state for a new batch; the (c, r) buffers are reused while the batch size stays the same, so
        the steady-state window loop allocates nothing."""
        dev = self.net.device
        for d in range(self.ndir):
            if not (self.stateful and d == 0):
                self.state[d] = None
            elif self.state[d] is not None and self.state[d][0].shape[0] == batch_size:
                for t_ in self.state[d]:
                    t_.zero_()
            else:
                self.state[d] = (torch.zeros(batch_size, self.N, device=dev),
                                 torch.zeros(batch_size, self.N, device=dev)) + \
                                ((torch.zeros(batch_size, self.P, device=dev),) if self.P else ())

    def _buf(self, name, rows, cols, zero=False):
        return self.net.arena.mat('L%d.%s' % (self.idx, name), rows, cols, zero)

    def prepare(self, train, T, B, mask, cond):
        """Everything of the forward pass that depends on weights, carried state and the conditioning input only, not on
        the layer below: the folded / packed recurrent matrices, the window's initial-state snapshot, the LHUC scale
        factors.  Network.forward() issues it for every layer on the side stream before the first scan starts
        (Network.prepare_ahead), so that between two scans the main stream carries only the input projection."""
        net = self.net; N = self.N; P = self.P
        Pout = P if P else N
        out = []
        for d, dp in enumerate(self.dirs):
            c = {}
            if P:
                Wf = self._buf('Wf%d' % d, N, 4 * N)
                ops.gemm(dp.p('Wp'), dp.p('Wh'), Wf)
            else:
                Wf = dp.p('Wh')
            c['wpk'] = ops.lstm_pack(Wf, N, 0, net.arena.get('L%d.wpk%d' % (self.idx, d),
                                                             _lib.lib().asr_lstm_pack_floats(N, 0)))
            if train:
                pm = _lib.lib().asr_lstm_bwd_pack_mode(N, self.ndir, self.ln_kind)
                c['wpkb'] = ops.lstm_pack(Wf, N, pm, net.arena.get('L%d.wpkb%d' % (self.idx, d),
                                                                   _lib.lib().asr_lstm_pack_floats(N, pm)))
            st = self.state[d]
            if st is not None:           # this window's initial state, kept for its backward pass
                c['c0'] = self._buf('c0_%d' % d, B, N); c['r0'] = self._buf('r0_%d' % d, B, N)
                c['c0'].copy_(st[0]); c['r0'].copy_(st[1])
                if P and len(st) > 2:
                    # hid_init is the projected state as the previous window left it (libs/lhuclstm_layers.py:313-319,
                    # :230-235), not r0 re-projected with a W_p that an optimiser step has moved since: the first
                    # step's recurrent pre-activation is h0 . W_h, handed to the scan as a0
                    c['h0'] = self._buf('h0_%d' % d, B, P); c['h0'].copy_(st[2])
                    c['a0'] = ops.gemm(c['h0'], dp.p('Wh'), self._buf('a0_%d' % d, B, 4 * N))
            if self.kind == 'lhuc':
                if net.seqsum is not None:                  # SeqSumLHUCLSTMLayer, :487-489
                    emb = net.seqsum.emb
                else:
                    emb = cond[:, 0, :].contiguous()        # IVectorLHUCLSTMLayer, :504-510
                acts = [emb]
                for i in range(self.npl):
                    a = self._buf('lh_a%d_%d' % (i, d), B, self.npu)
                    ops.gemm(acts[-1], dp.p('We%d' % i), a)
                    ops.bias_act(a, dp.p('be%d' % i), B, self.npu, 1)
                    acts.append(a)
                sf = self._buf('lh_sf%d' % d, B, Pout)
                ops.gemm(acts[-1], dp.p('We%d' % self.npl), sf, bias=dp.p('be%d' % self.npl))
                c['acts'] = acts; c['sf'] = sf
            out.append(c)
        return out

    def forward(self, X, T, B, mask, cond, train):
        net = self.net; N = self.N; P = self.P; rows = T * B
        Pout = P if P else N
        Y = self._buf('Y', rows, self.out_dim)
        desc = _lib.LstmScanDesc()
        desc.T, desc.B, desc.N, desc.ndir = T, B, N, self.ndir
        desc.ln_kind, desc.cell_ln, desc.grad_clip = self.ln_kind, self.cell_ln, self.clip
        desc.mask = mask.data_ptr(); desc.mask_stride_b = mask.stride(0); desc.mask_stride_t = mask.stride(1)
        self.ctx = ctx = dict(X=X, T=T, B=B, mask=mask, cond=cond, Y=Y, d=[])
        prep = net.take_prepared(self) or self.prepare(train, T, B, mask, cond)
        for d, dp in enumerate(self.dirs):
            c = prep[d]
            xin = self._buf('xin%d' % d, rows, 4 * N)
            if self.kind == 'icond_ln_prj':
                Tc = cond.shape[1]
                pre = self._buf('pre%d' % d, B * Tc, self.F)
                ops.gemm(cond.view(B * Tc, self.I), dp.p('Wc'), pre)
                ops.bias_act(pre, dp.p('bc'), B * Tc, self.F, 2)
                s = self._buf('s%d' % d, B, self.F)
                ops.masked_mean(pre, mask, B, Tc, self.F, s)
                al_in = self._buf('al_in%d' % d, B, 4 * N); al_hid = self._buf('al_hid%d' % d, B, 4 * N)
                bt = self._buf('bt%d' % d, B, 4 * N)
                ops.gemm(s, dp.p('Wai'), al_in, bias=dp.p('bai'))
                ops.gemm(s, dp.p('Wah'), al_hid, bias=dp.p('bah'))
                ops.gemm(s, dp.p('Wbt'), bt, bias=dp.p('bbt'))
                c.update(pre=pre, s=s, al_in=al_in, al_hid=al_hid, bt=bt)
            if self.ln_kind == 0:
                ops.gemm(X, dp.p('Wx'), xin, bias=dp.p('b'))
            else:
                ops.gemm(X, dp.p('Wx'), xin)
                xh = self._buf('xhat_in%d' % d, rows, 4 * N); rs = self._buf('rs_in%d' % d, rows, 1)
                if self.ln_kind == 1:
                    ops.ln_rows_fwd(xin, 4 * N, dp.p('a_in'), xin, beta=dp.p('b_in') if dp.has('b_in') else None,
                                    bias=dp.p('b'), xhat=xh, rstd=rs)
                else:
                    ops.ln_rows_fwd(xin, 4 * N, c['al_in'], xin, pre=c['bt'], pmod=B, amod=B, xhat=xh, rstd=rs)
                c.update(xhat_in=xh, rs_in=rs)
            Dd = desc.dir[d]
            Dd.reverse = 1 if self.reverse[d] else 0
            Dd.xin = xin.data_ptr(); Dd.wpack = c['wpk'].data_ptr()
            if self.ln_kind == 1:
                Dd.alpha_hid = dp.p('a_hid').data_ptr()
                Dd.beta_hid = dp.p('b_hid').data_ptr() if dp.has('b_hid') else None
            elif self.ln_kind == 2:
                Dd.alpha_hid = c['al_hid'].data_ptr()
            if self.cell_ln:
                Dd.alpha_c = dp.p('a_c').data_ptr(); Dd.beta_c = dp.p('b_c').data_ptr()
            if 'c0' in c:
                Dd.c0 = c['c0'].data_ptr(); Dd.r0 = c['r0'].data_ptr()
                if 'a0' in c:
                    Dd.a0 = c['a0'].data_ptr()
            direct = (not P) and self.kind != 'lhuc'       # scan writes straight into Y's column block
            if direct:
                R = Y[:, d * N:(d + 1) * N]
            else:
                R = self._buf('R%d' % d, rows, N)
            c['R'] = R
            Dd.r_out = R.data_ptr(); Dd.ld_r = R.stride(0)
            c['cst'] = self._buf('cst%d' % d, rows, N); Dd.cst = c['cst'].data_ptr()
            if train:
                c['gates'] = self._buf('gates%d' % d, rows, 4 * N); Dd.gates = c['gates'].data_ptr()
            if self.ln_kind:
                c['ahat'] = self._buf('ahat%d' % d, rows, 4 * N); c['rs_a'] = self._buf('rs_a%d' % d, rows, 1)
                Dd.ahat = c['ahat'].data_ptr(); Dd.rs_a = c['rs_a'].data_ptr()
            if self.cell_ln:
                c['chat'] = self._buf('chat%d' % d, rows, N); c['rs_c'] = self._buf('rs_c%d' % d, rows, 1)
                Dd.chat = c['chat'].data_ptr(); Dd.rs_c = c['rs_c'].data_ptr()
            ctx['d'].append(c)
        net.scan_lstm(self, desc, backward=False)
        if self.stateful and any(st is not None for st in self.state):
            with net.deferred():          # the next window's start: not on this window's critical path
                for d in range(self.ndir):
                    if self.state[d] is not None:               # libs/lhuclstm_layers.py:313-319
                        c = ctx['d'][d]
                        last = T - 1
                        if self.context:
                            last = (T if T < self.context else self.context) - 1
                        self.state[d][0].copy_(c['cst'][last * B:(last + 1) * B])
                        self.state[d][1].copy_(c['R'][last * B:(last + 1) * B])
        for d, dp in enumerate(self.dirs):
            c = ctx['d'][d]
            Yd = Y[:, d * Pout:(d + 1) * Pout]
            if P and self.kind != 'lhuc':
                ops.gemm(c['R'], dp.p('Wp'), Yd)
                hid = Yd
            elif self.kind == 'lhuc':
                if P:
                    hid = self._buf('hid%d' % d, rows, P)
                    ops.gemm(c['R'], dp.p('Wp'), hid)
                else:
                    hid = c['R']
                c['hid'] = hid
                ops.rowscale(hid, hid.stride(0), c['sf'], B, True, Yd, Yd.stride(0), rows, Pout)
            else:
                hid = None
            if P and self.stateful and self.state[d] is not None and len(self.state[d]) > 2:
                last = T - 1                   # the projected state, with THIS window's W_p (:313-319)
                if self.context:
                    last = (T if T < self.context else self.context) - 1
                self.state[d][2].copy_(hid[last * B:(last + 1) * B])
        return Y

    def backward(self, dY, need_dx=True):
        """BPTT of the pair.  Everything the layer below waits for (dR, the scan, the LN backward of
        the input side, dX) is issued first on the current stream; the parameter gradients are
        collected as closures and issued afterwards through net.deferred() (side stream, under the
        scan of the layer below)."""
        net = self.net; ctx = self.ctx; N = self.N; P = self.P
        X, T, B, mask, cond = ctx['X'], ctx['T'], ctx['B'], ctx['mask'], ctx['cond']
        rows = T * B
        Pout = P if P else N
        later = []
        early = []          # parameter gradients that need only dY: side stream, under this layer's own scan
        desc = _lib.LstmScanDesc()
        desc.T, desc.B, desc.N, desc.ndir = T, B, N, self.ndir
        desc.ln_kind, desc.cell_ln, desc.grad_clip = self.ln_kind, self.cell_ln, self.clip
        desc.mask = mask.data_ptr(); desc.mask_stride_b = mask.stride(0); desc.mask_stride_t = mask.stride(1)
        for d, dp in enumerate(self.dirs):
            c = ctx['d'][d]
            dYd = dY[:, d * Pout:(d + 1) * Pout]
            if self.kind == 'lhuc':
                sf = c['sf']; hid = c['hid']
                dhid = self._buf('lh_dhid%d' % d, rows, Pout)
                ops.rowscale(dYd, dYd.stride(0), sf, B, True, dhid, Pout, rows, Pout)
                early.append(lambda d=d, dp=dp, c=c, dYd=dYd: self._lhuc_param_grads(d, dp, c, dYd, rows, B, Pout))
                dYd = dhid
            if P:
                dR = self._buf('dR%d' % d, rows, N)
                ops.gemm(dYd, dp.p('Wp'), dR, tb=True)
                later.append(lambda dp=dp, c=c, dYd=dYd: ops.gemm(c['R'], dYd, dp.g('Wp'), ta=True))
            else:
                dR = dYd
            c['dR'] = dR
            Dd = desc.dir[d]
            Dd.reverse = 1 if self.reverse[d] else 0
            Dd.wpack = c['wpkb'].data_ptr()
            if self.ln_kind == 1:
                Dd.alpha_hid = dp.p('a_hid').data_ptr()
            elif self.ln_kind == 2:
                Dd.alpha_hid = c['al_hid'].data_ptr()
            if self.cell_ln:
                Dd.alpha_c = dp.p('a_c').data_ptr(); Dd.beta_c = dp.p('b_c').data_ptr()
            if 'c0' in c:
                Dd.c0 = c['c0'].data_ptr(); Dd.r0 = c['r0'].data_ptr()
            Dd.gates = c['gates'].data_ptr(); Dd.cst = c['cst'].data_ptr()
            if self.ln_kind:
                Dd.ahat = c['ahat'].data_ptr(); Dd.rs_a = c['rs_a'].data_ptr()
                c['dA'] = self._buf('dA%d' % d, rows, 4 * N); Dd.dA = c['dA'].data_ptr()
            if self.cell_ln:
                Dd.chat = c['chat'].data_ptr(); Dd.rs_c = c['rs_c'].data_ptr()
                c['dOut'] = self._buf('dOut%d' % d, rows, N); Dd.dOut = c['dOut'].data_ptr()
            Dd.dr_ext = dR.data_ptr(); Dd.ld_dr = dR.stride(0)
            c['dG'] = self._buf('dG%d' % d, rows, 4 * N); Dd.dG = c['dG'].data_ptr()
        if early:
            with net.deferred():
                for fn in early:
                    fn()
        net.scan_lstm(self, desc, backward=True)
        dX = self._buf('dX', rows, self.D) if need_dx else None
        for d, dp in enumerate(self.dirs):
            c = ctx['d'][d]
            dG = c['dG']
            if self.ln_kind == 0:
                dXp = dG
            else:
                dXp = self._buf('dXp%d' % d, rows, 4 * N)
                al = dp.p('a_in') if self.ln_kind == 1 else c['al_in']
                ops.ln_rows_bwd(dG, 4 * N, c['xhat_in'], c['rs_in'], al, dXp, amod=0 if self.ln_kind == 1 else B)
            c['dXp'] = dXp
            if need_dx:
                ops.gemm(dXp, dp.p('Wx'), dX, tb=True, beta=0.0 if d == 0 else 1.0)
            later.append(lambda d=d, dp=dp, c=c: self._param_grads(d, dp, c, X, cond, mask, rows, T, B))
        with net.deferred():
            for fn in later:
                fn()
        return dX

    def _lhuc_param_grads(self, d, dp, c, dYd, rows, B, Pout):
        sf = c['sf']; hid = c['hid']; acts = c['acts']
        dsf = self._buf('lh_dsf%d' % d, B, Pout)
        ops.colreduce(dYd, dsf, rows, Pout, y=hid, gmod=B, ldx=dYd.stride(0), ldy=hid.stride(0))
        ops.act_bwd(dsf, sf, B, Pout, 1)              # relu'(sf)
        ops.gemm(acts[-1], dsf, dp.g('We%d' % self.npl), ta=True)
        ops.colreduce(dsf, dp.g('be%d' % self.npl), B, Pout)
        da = dsf
        for i in range(self.npl - 1, -1, -1):
            dprev = self._buf('lh_da%d_%d' % (i, d), B, self.npu)
            ops.gemm(da, dp.p('We%d' % (i + 1)), dprev, tb=True)
            ops.act_bwd(dprev, acts[i + 1], B, self.npu, 1)
            ops.gemm(acts[i], dprev, dp.g('We%d' % i), ta=True)
            ops.colreduce(dprev, dp.g('be%d' % i), B, self.npu)
            da = dprev
        if self.net.seqsum is not None:       # the embedding is a function of the seq-summary parameters
            self.net.seqsum.accumulate(da, dp.p('We0'))

    def _param_grads(self, d, dp, c, X, cond, mask, rows, T, B):
        N = self.N; P = self.P
        dG = c['dG']; dA = c['dA'] if self.ln_kind else dG
        R = c['R']; dXp = c['dXp']
        # dW' = sum_t r_{prev(t)}^T dA_t
        dWf = self._buf('dWf%d' % d, N, 4 * N) if P else dp.g('Wh')
        n1 = (T - 1) * B
        if not self.reverse[d]:
            Ra, dAb, first = R[:n1], dA[B:], dA[:B]
        else:
            Ra, dAb, first = R[B:], dA[:n1], dA[n1:]
        ops.gemm(Ra, dAb, dWf, ta=True)
        if 'r0' in c and not P:
            ops.gemm(c['r0'], first, dWf, ta=True, beta=1.0)
        if P:
            ops.gemm(dp.p('Wp'), dWf, dp.g('Wh'), ta=True)
            ops.gemm(dWf, dp.p('Wh'), dp.g('Wp'), tb=True, beta=1.0)
            if 'r0' in c:
                # the carried hid_init = r0.W_p is a stored constant (libs/lhuclstm_layers.py:230-235): its
                # first-step term belongs to W_h alone, not to W_p through the folded matrix
                if 'h0' in c:
                    h0 = c['h0']
                else:                          # an injected (c, r) state (tests): h0 = r0 . W_p of the current weights
                    h0 = self._buf('h0_%d' % d, B, P)
                    ops.gemm(c['r0'], dp.p('Wp'), h0)
                ops.gemm(h0, first, dp.g('Wh'), ta=True, beta=1.0)
        if dp.has('b'):
            ops.colreduce(dG, dp.g('b'), rows, 4 * N)
        if self.cell_ln:
            ops.colreduce(c['dOut'], dp.g('a_c'), rows, N, y=c['chat'])
            ops.colreduce(c['dOut'], dp.g('b_c'), rows, N)
        if self.ln_kind == 1:
            ops.colreduce(dG, dp.g('a_in'), rows, 4 * N, y=c['xhat_in'])
            ops.colreduce(dG, dp.g('a_hid'), rows, 4 * N, y=c['ahat'])
            if dp.has('b_in'):
                ops.colreduce(dG, dp.g('b_in'), rows, 4 * N)
                ops.colreduce(dG, dp.g('b_hid'), rows, 4 * N)
        elif self.ln_kind == 2:
            F = self.F; Tc = cond.shape[1]
            dal_in = self._buf('dal_in%d' % d, B, 4 * N); dal_hid = self._buf('dal_hid%d' % d, B, 4 * N)
            dbt = self._buf('dbt%d' % d, B, 4 * N)
            ops.colreduce(dG, dal_in, rows, 4 * N, y=c['xhat_in'], gmod=B)
            ops.colreduce(dG, dal_hid, rows, 4 * N, y=c['ahat'], gmod=B)
            ops.colreduce(dXp, dbt, rows, 4 * N, gmod=B)
            s = c['s']
            ds = self._buf('ds%d' % d, B, F)
            for k, (nm, dv) in enumerate((('ai', dal_in), ('ah', dal_hid), ('bt', dbt))):
                ops.gemm(s, dv, dp.g('W' + nm), ta=True)
                ops.colreduce(dv, dp.g('b' + nm), B, 4 * N)
                ops.gemm(dv, dp.p('W' + nm), ds, tb=True, beta=0.0 if k == 0 else 1.0)
            dpre = self._buf('dpre%d' % d, B * Tc, F)
            ops.masked_mean_tanh_bwd(ds, mask, c['pre'], B, Tc, F, dpre)
            ops.gemm(cond.view(B * Tc, self.I), dpre, dp.g('Wc'), ta=True)
            ops.colreduce(dpre, dp.g('bc'), B * Tc, F)
        ops.gemm(X, dXp, dp.g('Wx'), ta=True)


def _init_tensor(rng, shape, init):
    """Reference initialisers (lasagne.init) on the host with numpy: Orthogonal (gain 1, SVD of a
    Gaussian), Constant, Normal(std), GlorotUniform."""
    if init == 'orthogonal':
        flat = (shape[0], int(np.prod(shape[1:])))
        a = rng.standard_normal(flat)
        u, _, v = np.linalg.svd(a, full_matrices=False)
        q = u if u.shape == flat else v
        return q.reshape(shape).astype(np.float32)
    if init == 'glorot_uniform':
        lim = np.sqrt(6.0 / (shape[0] + shape[1]))
        return rng.uniform(-lim, lim, shape).astype(np.float32)
    kind, v = init
    if kind == 'normal':
        return (rng.standard_normal(shape) * v).astype(np.float32)
    if kind == 'uniform':                              # lasagne.init.Uniform(v): U(-v, v)
        return rng.uniform(-v, v, shape).astype(np.float32)
    return np.full(shape, v, dtype=np.float32)


class SeqSumNet(object):
    """Sequence-summarising speaker network of libs/builder_utils.py:48-62: framewise tanh MLP on the
    network input, linear output, then SummarizingLayer = UNMASKED T.mean over time (:5-22) -> one
    embedding per utterance, shared by every SeqSumLHUCLSTMLayer (libs/lhuclstm_layers.py:473-489)."""
    def __init__(self, net, Din, num_units, num_layers, out_dim):
        self.net = net
        self.dims = [Din] + [num_units] * num_layers + [out_dim]
        self.nl = num_layers
        self.dp = DirParams(net.store, 'seqsum.')
        self.spec = []
        for i in range(num_layers + 1):
            self.dp.add('W%d' % i, (self.dims[i], self.dims[i + 1])); self.dp.add('b%d' % i, (self.dims[i + 1],))
            self.dp.map('seqsum_W_%d' % i, 'W%d' % i); self.dp.map('seqsum_b_%d' % i, 'b%d' % i)
            self.spec += [('seqsum_W_%d' % i, (self.dims[i], self.dims[i + 1]), 'glorot_uniform'),
                          ('seqsum_b_%d' % i, (self.dims[i + 1],), ('const', 0.0))]
        self.first = True

    def forward(self, X0, T, B):
        arena = self.net.arena; dp = self.dp; rows = T * B
        acts = [X0]
        for i in range(self.nl + 1):
            a = arena.mat('seqsum.a%d' % i, rows, self.dims[i + 1])
            ops.gemm(acts[-1], dp.p('W%d' % i), a)
            ops.bias_act(a, dp.p('b%d' % i), rows, self.dims[i + 1], 2 if i < self.nl else 0)
            acts.append(a)
        E = self.dims[-1]
        ones = arena.get('seqsum.ones', B * T)
        ones.fill_(1.0)
        self.emb = arena.mat('seqsum.emb', B, E)
        ops.seq_pool_fwd(acts[-1], E, B * E, ones, B, T, E, 2, self.emb)
        self.acts = acts; self.T, self.B = T, B; self.ones = ones
        self.demb = arena.mat('seqsum.demb', B, E)
        self.first = True
        return self.emb

    def accumulate(self, da, W0):
        """demb (+)= da . W0^T from one LHUC layer-direction (called in backward order)."""
        ops.gemm(da, W0, self.demb, tb=True, beta=0.0 if self.first else 1.0)
        self.first = False

    def backward(self):
        arena = self.net.arena; dp = self.dp; T, B = self.T, self.B; rows = T * B
        E = self.dims[-1]
        d = arena.mat('seqsum.d%d' % self.nl, rows, E)
        ops.seq_pool_bwd(self.demb, None, None, E, B * E, self.ones, B, T, E, 2, d)
        for i in range(self.nl, -1, -1):
            ops.gemm(self.acts[i], d, dp.g('W%d' % i), ta=True)
            ops.colreduce(d, dp.g('b%d' % i), rows, self.dims[i + 1])
            if i > 0:
                dprev = arena.mat('seqsum.d%d' % (i - 1), rows, self.dims[i])
                ops.gemm(d, dp.p('W%d' % i), dprev, tb=True)
                ops.act_bwd(dprev, self.acts[i], rows, self.dims[i], 2)
                d = dprev


class Network(object):
    """The stacked model + output layer.  Built by asr_amd.builders / asr_amd.gating_hyper_nets."""
    def __init__(self, device='cuda:0'):
        self.device = torch.device(device)
        self.store = ParamStore()
        self.arena = Arena(self.device)
        self.pairs = []
        self.pair_specs = []       # per pair: list (per direction) of [(ref_name, shape, init)]
        self.out = None
        self.seqsum = None           # SeqSumNet (build_deep_lhuclstm_seqsum*)
        self.ivector_concat = False
        self.uses_cond = False
        self.output_mode = 'softmax'      # what ff_fn returns: 'softmax' probabilities or 'logits'
        self._ws = {}
        self.finalized = False
        self.world = 1               # data-parallel world size (loss normaliser = global batch)
        self.grad_hook = None        # called as grad_hook(lo, hi) when flat grads [lo,hi) are final
        self.scan_events = None      # list of (kind, backward, rows, ndir, ev0, ev1) when profiling
        # parameter-gradient GEMMs of layer l run on a low-priority side stream under the BPTT scan
        # of layer l-1 (the scan is latency-bound and leaves the MFMA pipes idle)
        self.overlap = self.device.type == 'cuda' and os.environ.get('ASR_OVERLAP', '1') != '0'
        self.side = None
        # DropoutLayer(p) in front of every layer and of the output layer (models/gating_hyper_nets.py:276-307);
        # active when train=True (deterministic=False).  The mask is a pure function of (dropout_seed,
        # step, rank, layer): the backward pass regenerates it, nothing is stored.
        # time chunks per layer of the scans that support it (engine_hyper.HyperPair): the next layer's input
        # projection / this layer's dX run on the side stream under the remaining chunks.  Measured on cfg3
        # (DESIGN.md section 4): 1 / 2 / 4 / 8 chunks = 185.5 / 185.4 / 185.0 / 185.9 ms per step -- the GEMMs
        # that leave the critical path slow the latency-bound scans by as much as they save, so 1 is the default
        self.scan_chunks = int(os.environ.get('ASR_SCAN_CHUNKS', '1'))
        self.input_ready = False     # the output layer as a consumer of piecewise input (accumulate_input)
        self.dropout = 0.0
        self.dropout_seed = 0
        self.dropout_rank = 0
        self.dropout_step = 0
        self._drop_ids = []

    def set_precision(self, precision, allow_unbounded=False):
        """How the large GEMMs (input projections, output layer, dX, weight-gradient contractions) multiply; the
        recurrences, layer norms, softmax and the optimiser are fp32 in every mode.  Process-wide (asr_amd.ops).
          'f32x3' (default) fp32-grade products on the bf16 matrix cores from operands split into three bf16 parts;
          'f32'   fp32 MFMA throughout;
          'bf16x2' reduced precision with a bound on EVERY config: operands as two bf16 parts (16 significand bits),
                  3 bf16-MFMA products per fp32-grade one.  Against the float64 oracle (tests/test_gpu_bf16.py): log-
                  posteriors within 3e-4 (cfg3) / 1e-2 (the layer-norm stacks cfg2, cfg4), the whole gradient within
                  1e-3 / 2e-2 of its norm; cfg3 step 5 % faster than the default, the LN stacks 2 %;
          'bf16'  the optional mode of BASELINE.json's north_star: operands rounded to bf16, fp32 accumulate,
                  with the measured bound of tests/test_gpu_bf16.py.  REFUSED for networks with layer norm inside
                  the recurrence (LN-LSTM(P), cond-LN, ProjectionHyper, the scaling layer's layer-norm option): no
                  bound can be stated there -- measured on cfg2, |d log p| up to 0.59 and a gradient error of 0.93
                  of the gradient's norm (the LN stack amplifies even fp32 rounding to 1e-3) -- so the mode raises
                  instead of silently training on noise; allow_unbounded=True overrides (measurements only)."""
        if precision not in ops.GEMM_TERMS:
            raise ValueError('precision must be one of %s' % sorted(ops.GEMM_TERMS))
        if precision == 'bf16' and not allow_unbounded:
            ln = [p.idx for p in self.pairs if getattr(p, 'ln_kind', 0) or getattr(p, 'cell_ln', 0) or
                  getattr(p, 'ln', False) is True or getattr(p, 'kind', '') == 'prjhyper']
            if ln:
                raise _lib.AsrError('set_precision(\'bf16\'): layers %s normalise inside the recurrence; the bf16 mode has '
                                    'no stated bound for such stacks (tests/test_gpu_bf16.py) and is refused -- use '
                                    'the default f32x3' % ln)
        ops.PRECISION = precision
        self.precision = precision

    # ---- construction -------------------------------------------------------------------------
    def add_pair(self, pair, spec_per_dir):
        self.pairs.append(pair)
        self.pair_specs.append(spec_per_dir)
        return pair

    def set_output(self, K, C_, mode):
        self.out = DirParams(self.store, 'out.')
        self.out.add('W', (K, C_)); self.out.add('b', (C_,))
        self.out.map('W', 'W'); self.out.map('b', 'b')
        self.out_spec = specs.dense_params(K, C_)
        self.K_top = K; self.C = C_
        self.output_mode = mode

    def finalize(self, seed=0):
        self.store.materialize(self.device)
        rng = np.random.default_rng(seed)
        if self.seqsum is not None:
            vals = {n: torch.from_numpy(_init_tensor(rng, s, i)).to(self.device) for n, s, i in self.seqsum.spec}
            self.seqsum.dp.load_ref(vals)
        for pair, sp in zip(self.pairs, self.pair_specs):
            for dp, spec in zip(pair.dirs, sp):
                vals = {n: torch.from_numpy(_init_tensor(rng, s, i)).to(self.device) for n, s, i in spec}
                dp.load_ref(vals)
        vals = {n: torch.from_numpy(_init_tensor(rng, s, i)).to(self.device) for n, s, i in self.out_spec}
        self.out.load_ref(vals)
        self.m = torch.zeros_like(self.store.p)
        self.v = torch.zeros_like(self.store.p)
        self.t_prev = 0
        self.dropout_seed = (int(seed) * 0x9E3779B97F4A7C15 + 0x2545F4914F6CDD1D) % (1 << 64)
        self.scal = torch.zeros(16, dtype=torch.float32, device=self.device)
        self.adam_ws = torch.zeros(1024, dtype=torch.float32, device=self.device)
        self.finalized = True
        return self

    # ---- reference-compatible parameter lists -------------------------------------------------
    def _all_dirparams(self):
        out = [self.seqsum.dp] if self.seqsum is not None else []   # incomings-first order: SURVEY 8b.3
        for pair in self.pairs:
            out.extend(pair.dirs)
        out.append(self.out)
        return out

    def param_names_shapes(self):
        res = []
        for dp in self._all_dirparams():
            for ref_name, name, kind, arg in dp.refmap:
                E = dp.p(name)
                v = _PermView(E).to_ref() if kind == 'perm4' else dp._view(E, kind, arg)
                res.append((ref_name, tuple(v.shape)))
        return res

    def get_all_params(self, trainable=True):
        """= lasagne.layers.get_all_params(network, trainable=True): handles with the shared-variable
        protocol the reference uses on them (.name, .get_value(), .set_value(v)), in the
        reference's order."""
        out = []
        for dp in self._all_dirparams():
            for i in range(len(dp.refmap)):
                out.append(ParamHandle(dp, i))
        return out

    def get_all_param_values(self, grads=False):
        """= lasagne.layers.get_all_param_values(network, trainable=True): list of numpy arrays in
        the reference's order (libs/utils.py:80-85 pickles exactly this list)."""
        return [v for dp in self._all_dirparams() for _, v in dp.dump_ref(grads)]

    def get_all_grad_values(self):
        return self.get_all_param_values(grads=True)

    def set_all_param_values(self, values):
        """libs/lasagne_libs/utils.py:2-14: same count / shape checks, same ValueError text."""
        ns = self.param_names_shapes()
        if len(ns) != len(values):
            raise ValueError("mismatch: got %d values to set %d parameters" % (len(values), len(ns)))
        for (n, shp), v in zip(ns, values):
            if tuple(np.shape(v)) != shp:
                raise ValueError("mismatch: parameter has shape %r but value to set has shape %r"
                                 % (shp, tuple(np.shape(v))))
        k = 0
        for dp in self._all_dirparams():
            vals = {}
            for ref_name, _, _, _ in dp.refmap:
                vals[ref_name] = torch.from_numpy(np.ascontiguousarray(values[k], dtype=np.float32)).to(self.device)
                k += 1
            dp.load_ref(vals)

    def describe(self):
        """Plain-data description of the wiring (consumed by the test oracle)."""
        layers = []
        for pair in self.pairs:
            layers.append(pair.describe())
        d = dict(layers=layers, output=self.output_mode, ivector_concat=self.ivector_concat,
                 cond=self.uses_cond)
        if self.seqsum is not None:
            d['seqsum'] = dict(num_layers=self.seqsum.nl, params=[r[0] for r in self.seqsum.dp.refmap])
        return d

    def tbptt_layers(self):
        return [p for p in self.pairs if getattr(p, 'stateful', False)]

    @contextlib.contextmanager
    def deferred(self):
        """Work issued inside runs after everything enqueued so far on the current stream, but off the
        critical path (side stream); Network.backward() joins it before returning."""
        if not self.overlap:
            yield
            return
        if self.side is None:
            try:
                self.side = torch.cuda.Stream(device=self.device, priority=1)
            except Exception:
                self.side = torch.cuda.Stream(device=self.device)
        ev = torch.cuda.Event()
        ev.record()
        self.side.wait_event(ev)
        with torch.cuda.stream(self.side):
            yield

    def join_deferred(self):
        if self.side is not None:
            torch.cuda.current_stream().wait_stream(self.side)

    def prepare_ahead(self, train, T, B, mask, cond):
        """Issues every layer's weight-only forward work (pair.prepare) on the side stream, behind everything enqueued
        so far (the optimiser step that produced the weights), before the first layer's scan starts.  Measured on a
        TBPTT window of cfg5 under rocprofv3: ~75 us of tiny launches per layer (LHUC scale MLP, packs, state snapshot)
        sat between two 320 us scans on the main stream; cfg3: the 24 fold products + packs of a HyperLSTM layer."""
        if not self.overlap:
            return
        todo = [p for p in self.pairs if hasattr(p, 'prepare')]
        if not todo:
            return
        with self.deferred():
            for p in todo:
                p._prepared = p.prepare(train, T, B, mask, cond)
                p._prepared_ev = torch.cuda.Event()
                p._prepared_ev.record()

    def take_prepared(self, pair):
        """pair.forward(): what prepare_ahead() left for this pair (the main stream then waits for it), or None."""
        prep = getattr(pair, '_prepared', None)
        if prep is None:
            return None
        pair._prepared = None
        torch.cuda.current_stream().wait_event(pair._prepared_ev)
        return prep

    # ---- scan launch helpers ------------------------------------------------------------------
    def _workspace(self, key, nbytes):
        ws = self._ws.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.zeros(int(nbytes) + 256, dtype=torch.uint8, device=self.device)
            self._ws[key] = ws
        return ws

    def regularizable_ranges(self, is_reg):
        """Flat-buffer ranges [lo, hi) of the engine tensors all of whose reference parameters are
        regularizable (lasagne's `regularizable` tag, decided by `is_reg(ref_name)`).  An engine tensor
        that mixes both kinds cannot be split into ranges and is refused."""
        offs = {key: (off, off + (int(np.prod(shape)) + 3) // 4 * 4) for key, shape, off in self.store.entries}
        out = []
        for dp in self._all_dirparams():
            flags = {}
            for ref_name, name, kind, arg in dp.refmap:
                flags.setdefault(name, set()).add(bool(is_reg(ref_name)))
            for name, fl in flags.items():
                if len(fl) > 1:
                    raise NotImplementedError('asr_amd: engine tensor %s%s mixes regularizable and '
                                              'non-regularizable reference parameters' % (dp.prefix, name))
                if True in fl:
                    out.append(offs[dp.keys[name]])
        return out

    def add_weight_decay(self, coef, is_reg):
        """grad += 2*coef*param on the regularizable tensors: the gradient of lasagne's
        apply_penalty(params, l2) * coef (scripts/dln/ted/exp_prj_lstm_dln.py:129-131).  Runs as the sgd
        form of asr_opt_step with the roles swapped (g <- g - (-2 coef) p)."""
        if not hasattr(self, '_wd_ranges'):
            self._wd_ranges = self.regularizable_ranges(is_reg)
        for lo, hi in self._wd_ranges:
            ops.opt_step(0, self.store.g[lo:hi], self.store.p[lo:hi], None, None, -2.0 * float(coef), 0.0, 0.0, 0.0)

    def param_range(self, prefix):
        offs = [(off, off + (int(np.prod(shape)) + 3) // 4 * 4) for key, shape, off in self.store.entries
                if key.startswith(prefix)]
        return min(o[0] for o in offs), max(o[1] for o in offs)

    def _timed(self, kind, backward, desc, call):
        if self.scan_events is None:
            return call()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        r = call()
        if r != 0:                  # a refused launch (the caller may retry per direction): nothing to time
            return r
        e1.record()
        steps = (getattr(desc, 'step1', 0) or desc.T) - getattr(desc, 'step0', 0)
        self.scan_events.append((kind, bool(backward), steps * desc.B, desc.ndir, e0, e1))
        return r

    def scan_lstm(self, pair, desc, backward, key=None):
        L = _lib.lib()
        what = 'asr_lstm_scan_%s' % ('bwd' if backward else 'fwd')
        key = key or ('lstm', pair.idx, backward)
        need = L.asr_lstm_scan_ws_bytes(desc.T, desc.B, desc.N, desc.ndir, int(backward))
        ws = self._workspace(key, need)
        fn = L.asr_lstm_scan_bwd if backward else L.asr_lstm_scan_fwd
        rc = self._timed('lstm', backward, desc, lambda: fn(C.byref(desc), ws.data_ptr(), ws.numel(),
                                                            torch.cuda.current_stream().cuda_stream))
        if rc == -5 and desc.ndir == 2:        # ASR_ERR_LIMIT
            # the library refuses a grid or an LDS share that cannot be co-resident before it launches anything; the two
            # directions of a layer are independent, so a layer too wide for one launch gets one launch per direction
            for d in range(2):
                one = type(desc)()
                C.memmove(C.byref(one), C.byref(desc), C.sizeof(desc))
                one.ndir = 1
                one.dir[0] = desc.dir[d]
                self.scan_lstm(pair, one, backward, key=('lstm', pair.idx, backward, d))
            return
        if rc == -5:
            raise _lib.AsrError('%s failed: ASR_ERR_LIMIT (-5): %d units are more than the persistent scan holds (at most '
                                '256 co-resident workgroups of 4, 8 or 16 units per direction: 4096 units, a multiple of 16 '
                                'above 2048; up to ~1280 units with the weights resident in LDS, streamed from L2 beyond; '
                                'layers with peepholes / learn_init / cell_out only while the weights are resident)'
                                % (what, desc.N))
        check(rc, what)
        self._last_groups = getattr(self, '_last_groups', {})
        self._last_groups[key] = desc.ndir * ((desc.B + 15) // 16)

    def scan_hyper(self, pair, desc, backward, key=None):
        L = _lib.lib()
        if desc.ndir == 2 and 128 < desc.N // 4 <= 256:
            # one workgroup per 4 units and 256 co-resident workgroups: above 512 units the two directions of a
            # bidirectional layer do not fit one launch; they are independent, so each gets a launch of its own
            for d in range(2):
                one = type(desc)()
                C.memmove(C.byref(one), C.byref(desc), C.sizeof(desc))
                one.ndir = 1
                one.dir[0] = desc.dir[d]
                self.scan_hyper(pair, one, backward, key=('hyper', pair.idx, backward, d))
            return
        key = key or ('hyper', pair.idx, backward)
        if desc.N // 4 > 256:
            raise _lib.AsrError('asr_hyper_scan: %d units are more than the resident-weight scan holds (one workgroup per 4 '
                                'units, at most 256 co-resident workgroups per direction: 1024 units)' % desc.N)
        need = L.asr_hyper_scan_ws_bytes(desc.T, desc.B, desc.N, desc.H, desc.ndir, int(backward))
        ws = self._workspace(key, need)
        fn = L.asr_hyper_scan_bwd if backward else L.asr_hyper_scan_fwd
        check(self._timed('hyper', backward, desc, lambda: fn(C.byref(desc), ws.data_ptr(), ws.numel(),
                                                              torch.cuda.current_stream().cuda_stream)),
              'asr_hyper_scan_%s' % ('bwd' if backward else 'fwd'))
        self._last_groups = getattr(self, '_last_groups', {})
        self._last_groups[key] = desc.ndir * ((desc.B + 7) // 8)

    def check_status(self):
        """Call after a host sync: raises if any persistent scan gave up on a bounded wait."""
        torch.cuda.synchronize(self.device)
        for key, ng in getattr(self, '_last_groups', {}).items():
            check(_lib.lib().asr_scan_status(self._ws[key].data_ptr(), ng), 'persistent scan %r' % (key,))

    # ---- forward / loss / backward ------------------------------------------------------------
    def _prep_input(self, x, cond):
        B, T, D = x.shape
        Din = D + (cond.shape[2] if self.ivector_concat else 0)
        X0 = self.arena.mat('X0', T * B, Din)
        ops.swap01(x, X0, B, T, D, D, Din)
        if self.ivector_concat:                      # libs/deep_lstm_builder.py:14-16
            I = cond.shape[2]
            ops.swap01(cond, X0[:, D:], B, T, I, I, Din)
        return X0

    def forward(self, x, mask, cond=None, train=False):
        """x (B,T,D) float32 cuda, mask (B,T) float32, cond (B,T,I).  Returns time-major logits
        (T*B, C)."""
        assert self.finalized
        B, T, D = x.shape
        x = x.contiguous(); mask = mask.contiguous()
        if cond is not None:
            cond = cond.contiguous()
        self.B, self.T = B, T
        act = self._prep_input(x, cond)
        if self.seqsum is not None:
            self.seqsum.forward(act, T, B)
        drop = bool(train) and self.dropout > 0
        if drop:
            self.dropout_step += 1
        self._drop_ids = []
        self.prepare_ahead(train, T, B, mask, cond)
        for l, pair in enumerate(self.pairs):
            if drop:
                act = self._dropout_fwd(l, act)
            nxt = self.pairs[l + 1] if l + 1 < len(self.pairs) else self
            if self.overlap and not drop and hasattr(pair, 'chunked_forward') and hasattr(nxt, 'accumulate_input'):
                act = pair.forward(act, T, B, mask, cond, train, consumer=nxt)
            else:
                act = pair.forward(act, T, B, mask, cond, train)
        if drop:
            act = self._dropout_fwd(len(self.pairs), act)
        self.top = act
        logits = self.arena.mat('logits', T * B, self.C)
        if self.input_ready:                     # already projected piecewise under the top layer's scan
            self.input_ready = False
        else:
            ops.gemm(act, self.out.p('W'), logits, bias=self.out.p('b'))   # libs/builder_utils.py:44-45
        self.logits = logits
        self.join_deferred()       # carried-state copies of stateful layers (side stream) before anything reads the state
        return logits

    # the output layer as the consumer of a chunked top layer (same protocol as HyperPair.begin_input)
    def begin_input(self, T, B):
        self._logits_in = self.arena.mat('logits', T * B, self.C)
        self.input_ready = True

    def accumulate_input(self, Xpart, r0, r1, k0, first):
        ops.gemm(Xpart, self.out.p('W')[k0:k0 + Xpart.shape[1]], self._logits_in[r0:r1], beta=0.0 if first else 1.0,
                 bias=self.out.p('b') if first else None)

    def _dropout_fwd(self, l, X):
        sid = (self.dropout_step << 20) | ((self.dropout_rank & 0xfff) << 8) | l
        self._drop_ids.append(sid)
        Xd = self.arena.mat('drop%d' % l, X.shape[0], X.shape[1])
        return ops.dropout(X, Xd, X.shape[0], X.shape[1], self.dropout, self.dropout_seed, sid)

    def _dropout_bwd(self, l, dX):
        """dX: gradient wrt the dropped input of layer l (l = len(pairs): the output layer), scaled in
        place by the regenerated mask."""
        if l < len(self._drop_ids):
            ops.dropout(dX, dX, dX.shape[0], dX.shape[1], self.dropout, self.dropout_seed, self._drop_ids[l])
        return dX

    def dropout_masks(self):
        """The scaled masks keep/(1-p) of the last train-mode forward, one (T*B, width) matrix per
        DropoutLayer (tests hand them to the oracle)."""
        out = []
        for l, sid in enumerate(self._drop_ids):
            n = self.arena.bufs['drop%d' % l].numel() // (self.T * self.B)
            ones = torch.ones(self.T * self.B, n, device=self.device)
            out.append(ops.dropout(ones, ones, ones.shape[0], n, self.dropout, self.dropout_seed, sid))
        return out

    def posteriors(self, x, mask, cond=None, log=False):
        """ff_fn body (libs/comp_graph_utils.py:321-333): (B,T,C) probabilities, or the raw logits for
        gating_hyper_nets models; log=True gives the max-subtracted log-softmax of
        bin/forward_ln_lstmp.py:55-58."""
        logits = self.forward(x, mask, cond, train=False)
        B, T = self.B, self.T
        out = torch.empty(B, T, self.C, dtype=torch.float32, device=self.device)
        if self.output_mode == 'logits' and not log:
            ops.swap01(logits, out, T, B, self.C)
        else:
            ops.softmax_ce(logits, self.C, B, T, out_mode=1 if log else 2, out=out)
        return out

    def loss(self, x, mask, target, target_mask, cond=None, train=True, context=0, delay=0):
        """compute_loss (libs/comp_graph_utils.py:29-45) / compute_loss_tbptt (:47-66, context form).
        Leaves dlogits (scaled by 1/n_batch, libs/comp_graph_utils.py:40) ready for backward()."""
        logits = self.forward(x, mask, cond, train=train)
        B, T = self.B, self.T
        rows = T * B
        tm = target_mask.contiguous()
        cut = context if (context and context < T) else 0
        if cut or delay:                           # scored frames only; an arena copy, no allocation
            eff = self.arena.get('tmask_eff', B * T).view(B, T)
            eff.copy_(tm)
            if cut:                                # context_tbptt(): o[:, :context], libs/comp_graph_utils.py:26-27
                eff[:, cut:].zero_()
            if delay:                              # delayed(): o[:, delay:], libs/comp_graph_utils.py:18-24
                eff[:, :delay].zero_()
            tm = eff
        target = target.contiguous()
        dl = self.arena.mat('dlogits', rows, self.C) if train else None
        row_ce = self.arena.get('row_ce', rows); row_ok = self.arena.get('row_ok', rows)
        pred = torch.empty(B, T, dtype=torch.int32, device=self.device)
        ops.softmax_ce(logits, self.C, B, T, target=target, tmask=tm, scale=1.0 / (B * self.world), dlogits=dl,
                       row_ce=row_ce, row_ok=row_ok, pred=pred)
        ops.reduce_sum(row_ce, rows, self.scal[0:1])
        ops.reduce_sum(row_ok, rows, self.scal[1:2])
        self.dlogits = dl
        if context and context < T:
            pred = pred[:, :context]
        if delay:
            pred = pred[:, delay:]
        return self.scal[0:1], self.scal[1:2], pred

    def backward(self):
        """theano.grad(ce_cost, params) (libs/comp_graph_utils.py:73-75) into the flat grad buffer."""
        dl = self.dlogits
        dY = self.arena.mat('dtop', dl.shape[0], self.K_top)
        ops.gemm(dl, self.out.p('W'), dY, tb=True)
        self._dropout_bwd(len(self.pairs), dY)
        with self.deferred():
            ops.gemm(self.top, dl, self.out.g('W'), ta=True)
            ops.colreduce(dl, self.out.g('b'), dl.shape[0], self.C)
            if self.grad_hook:
                self.grad_hook(*self.param_range('out.'))
        for i in range(len(self.pairs) - 1, -1, -1):
            dY = self.pairs[i].backward(dY, need_dx=(i > 0))
            if i > 0:
                self._dropout_bwd(i, dY)
            if self.grad_hook:
                with self.deferred():          # after this layer's (deferred) parameter gradients
                    self.grad_hook(*self.param_range('L%d.' % i))
        if self.seqsum is not None:
            with self.deferred():
                self.seqsum.backward()
                if self.grad_hook:
                    self.grad_hook(*self.param_range('seqsum.'))
        self.join_deferred()

    # ---- gradient accumulation over micro-batches: a batch whose saved BPTT state exceeds HBM (cfg3: 0.75 MB per
    # (t, b) row over 4 layer pairs, i.e. B = 128 x T = 2900 would need 276 GB) runs as several loss()/backward()
    # passes; the caller sets `world` to the number of equal parts so that every part is scaled by 1/B_total
    # (libs/comp_graph_utils.py:40), and the sum is what theano.grad of the whole batch gives.
    def grad_accumulate(self, first):
        if getattr(self, '_gacc', None) is None:
            self._gacc = torch.empty_like(self.store.g)
        if first:
            self._gacc.copy_(self.store.g)
        else:                                        # sgd step with lr = -1: acc <- acc + g, exact
            ops.opt_step(0, self._gacc, self.store.g, None, None, -1.0, 0.0, 0.0, 0.0)

    def grad_accumulated(self):
        self.store.g.copy_(self._gacc)

    def adam_update(self, lr, b1=0.9, b2=0.999, eps=1e-8, want_sqrt=True):
        """libs/lasagne_libs/updates.py:587-617 on the flat buffer + grads_norm
        (libs/comp_graph_utils.py:77; TBPTT trainer returns the un-rooted sum, :111)."""
        self.t_prev += 1
        t = self.t_prev
        a_t = lr * np.sqrt(1.0 - b2 ** t) / (1.0 - b1 ** t)
        ops.adam_step(self.store.p, self.store.g, self.m, self.v, float(a_t), b1, b2, eps,
                      self.scal[2:3], want_sqrt, self.adam_ws)
        return self.scal[2:3]

    def opt_update(self, kind, lr, h1=0.0, h2=0.0, eps=0.0, max_grad_norm=0.0, norm_eps=1e-7, want_sqrt=True):
        """One step of optimiser `kind` (asr_opt_step numbering) on the flat buffers; with
        max_grad_norm > 0 the gradients are first scaled by total_norm_constraint's multiplier
        (libs/lasagne_libs/updates.py:781-855), all on the device.  Returns the gradient norm (device
        scalar) BEFORE scaling, which is what return_norm=True hands back."""
        n = self.store.p.numel()
        if not hasattr(self, 'opt_state'):
            self.opt_state = {}
        st = self.opt_state.get(kind)
        if st is None:
            st = [torch.zeros_like(self.store.p) if kind >= 1 else None,
                  torch.zeros_like(self.store.p) if kind >= 5 else None]
            if kind == 7:
                st = [self.m, self.v]
            self.opt_state[kind] = st
        ops.sqsum(self.store.g, self.scal[2:3], True, self.adam_ws)
        gscale = None
        if max_grad_norm and max_grad_norm > 0:
            ops.norm_clip_mult(self.scal[2:3], float(max_grad_norm), float(norm_eps), self.scal[4:5])
            gscale = self.scal[4:5]
        ops.opt_step(kind, self.store.p, self.store.g, st[0], st[1], float(lr), float(h1), float(h2), float(eps), gscale)
        if not want_sqrt:
            ops.sqsum(self.store.g, self.scal[2:3], False, self.adam_ws)
        return self.scal[2:3]

    def grads_norm(self, want_sqrt=True):
        ops.sqsum(self.store.g, self.scal[2:3], want_sqrt, self.adam_ws)
        return self.scal[2:3]

"""ScalingHyperLSTMLayer pair (libs/lasagne_libs/hyper_layers.py:525-1205, use_peepholes=False,
use_layer_norm=False) on the persistent HyperLSTM scan kernel (asr_amd/csrc/scan_hyper.hip).

The layer is the HyperLSTM recurrence with the scale maps given directly instead of through the
low-rank eq.11/12 product:
  inner cell  = the kernel's hyper cell: gates = x.W_inner_in + b_inner (hoisted, `xhproj`)
                + h_in.W_inner_hid (`Whh`) + h_out.W_outer_hid_to_inner (`Whx`)         (:973-975)
  scale maps  = h_in.[S_hid | S_in | S_bias] + [s_hid | s_in | s_bias] = the kernel's M (H+1,12N),
                tau order (recurrent term, input term, bias)                            (:1030-1040)
  outcell clip on both cells (:1016-1022,:1083-1089) -> asr_hyper_scan_desc.clip_outcell.
The kernel needs num_inner_units % 4 == 0; other widths (the reference scripts use 250) are zero
padded to Hp: padded inner units have zero weights in and out, stay at h = c = 0 and receive zero
gradients, so results and the reference-shaped parameter views are unaffected."""
import torch

from . import _lib, ops, specs
from ._lib import check
from .engine import DirParams

G4 = specs.GATES_REC
_TAU = (('hid', 0), ('in', 1), ('bias', 2))


def _mk_scaling(store, prefix, D, N, H, Hp):
    dp = DirParams(store, prefix)
    dp.add('Whh', (Hp, 4 * Hp)); dp.add('Wxh', (N + D, 4 * Hp)); dp.add('bh', (4 * Hp,))
    dp.add('M', (Hp + 1, 12 * N))
    dp.add('Wh', (N, 4 * N)); dp.add('Wx', (D, 4 * N))
    gc = lambda r0, r1, nc, g: (lambda E: E[r0:r1].view(r1 - r0, -1, 4)[:, :nc, g])
    mw = lambda tau, g: (lambda E: E[:H].view(H, 3, N, 4)[:, tau, :, g])
    mb = lambda tau, g: (lambda E: E[Hp].view(3, N, 4)[tau, :, g])
    # reference add_param order: hyper_layers.py:570-600 (inner), :670-760 (outer), per gate
    for g, gn in enumerate(G4):
        dp.map('W_inner_in_to_inner_' + gn, 'Wxh', 'fn', gc(N, N + D, H, g))
        dp.map('W_inner_hid_to_inner_' + gn, 'Whh', 'fn', gc(0, H, H, g))
        dp.map('W_outer_hid_to_inner_' + gn, 'Wxh', 'fn', gc(0, N, H, g))
        dp.map('b_inner_' + gn, 'bh', 'fn', (lambda g: (lambda E: E.view(-1, 4)[:H, g]))(g))
    for g, gn in enumerate(G4):
        dp.map('W_outer_in_to_outer_' + gn, 'Wx', 'gatecols', g)
        dp.map('W_inner_hid_to_outer_in_' + gn, 'M', 'fn', mw(1, g))
        dp.map('b_inner_hid_to_outer_in_' + gn, 'M', 'fn', mb(1, g))
        dp.map('W_outer_hid_to_outer_' + gn, 'Wh', 'gatecols', g)
        dp.map('W_inner_hid_to_outer_hid_' + gn, 'M', 'fn', mw(0, g))
        dp.map('b_inner_hid_to_outer_hid_' + gn, 'M', 'fn', mb(0, g))
        dp.map('W_inner_hid_to_outer_bias_' + gn, 'M', 'fn', mw(2, g))
        dp.map('b_inner_hid_to_outer_bias_' + gn, 'M', 'fn', mb(2, g))
    return dp


class ScalingPair(object):
    kind = 'scaling'
    stateful = False

    def __init__(self, net, idx, D, N, H, clip, ndir):
        self.net = net; self.idx = idx
        self.D, self.N, self.H = D, N, H
        self.Hp = (H + 3) // 4 * 4
        self.clip = float(clip); self.ndir = ndir
        self.out_dim = ndir * N
        self.dirs = [_mk_scaling(net.store, 'L%d.%d.' % (idx, d), D, N, H, self.Hp) for d in range(ndir)]
        self.reverse = [False, True][:ndir]

    def describe(self):
        out = {}
        for d, nm in enumerate(('fwd', 'bwd')):
            if d >= self.ndir:
                out[nm] = None
                continue
            out[nm] = dict(kind='scaling',
                           args=dict(num_inner_units=self.H, num_outer_units=self.N,
                                     grad_clipping=self.clip, backwards=self.reverse[d]),
                           params=[r[0] for r in self.dirs[d].refmap])
        return out

    def _buf(self, name, rows, cols):
        return self.net.arena.mat('L%d.%s' % (self.idx, name), rows, cols)

    def _desc(self, T, B, mask):
        desc = _lib.HyperScanDesc()
        desc.T, desc.B, desc.N, desc.H, desc.ndir = T, B, self.N, self.Hp, self.ndir
        desc.grad_clip = self.clip
        desc.clip_outcell = 1
        desc.mask = mask.data_ptr(); desc.mask_stride_b = mask.stride(0); desc.mask_stride_t = mask.stride(1)
        return desc

    def inner_hid(self, d):
        """(T*B, num_inner_units) time-major inner hidden sequence of direction d of the last forward
        (what get_inner_hid=True exposes, models/gating_hyper_nets.py:100-113)."""
        return self.ctx['d'][d]['Hh'][:, :self.H]

    def forward(self, X, T, B, mask, cond, train):
        net = self.net; N, H = self.N, self.Hp
        rows = T * B
        L = _lib.lib()
        Y = self._buf('Y', rows, self.out_dim)
        desc = self._desc(T, B, mask)
        self.ctx = ctx = dict(X=X, T=T, B=B, mask=mask, Y=Y, d=[])
        for d, dp in enumerate(self.dirs):
            c = {}
            c['xproj'] = ops.gemm(X, dp.p('Wx'), self._buf('xproj%d' % d, rows, 4 * N))
            c['xhproj'] = ops.gemm(X, dp.p('Wxh')[N:], self._buf('xhproj%d' % d, rows, 4 * H), bias=dp.p('bh'))
            Mi = dp.p('M')
            for mode, nm in ((0, 'wpk'), (1, 'wpkb')):
                if mode == 1 and not train:
                    continue
                n = L.asr_hyper_pack_floats(N, H, mode)
                out = net.arena.get('L%d.%s%d' % (self.idx, nm, d), n)
                check(L.asr_hyper_pack_w(dp.p('Wh').data_ptr(), dp.p('Wxh').data_ptr(), dp.p('Whh').data_ptr(),
                                         Mi.data_ptr(), N, H, mode, out.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), 'asr_hyper_pack_w')
                c[nm] = out
            Dd = desc.dir[d]
            Dd.reverse = 1 if self.reverse[d] else 0
            Dd.xproj = c['xproj'].data_ptr(); Dd.xhproj = c['xhproj'].data_ptr()
            Dd.wpack = c['wpk'].data_ptr(); Dd.cvec = Mi[H].data_ptr()
            R = Y[:, d * N:(d + 1) * N]
            c['R'] = R
            Dd.r_out = R.data_ptr(); Dd.ld_r = R.stride(0)
            c['Hh'] = self._buf('Hh%d' % d, rows, H); Dd.hh_out = c['Hh'].data_ptr()
            if train:
                c['gates'] = self._buf('gates%d' % d, rows, 4 * N); Dd.gates = c['gates'].data_ptr()
                c['cst'] = self._buf('cst%d' % d, rows, N); Dd.cst = c['cst'].data_ptr()
                c['hgates'] = self._buf('hgates%d' % d, rows, 4 * H); Dd.hgates = c['hgates'].data_ptr()
                c['hcst'] = self._buf('hcst%d' % d, rows, H); Dd.hcst = c['hcst'].data_ptr()
                c['aD'] = self._buf('aD%d' % d, rows, 12 * N); Dd.aD = c['aD'].data_ptr()
            ctx['d'].append(c)
        net.scan_hyper(self, desc, backward=False)
        return Y

    def backward(self, dY, need_dx=True):
        net = self.net; ctx = self.ctx; N, H = self.N, self.Hp
        X, T, B, mask = ctx['X'], ctx['T'], ctx['B'], ctx['mask']
        rows = T * B
        desc = self._desc(T, B, mask)
        for d, dp in enumerate(self.dirs):
            c = ctx['d'][d]
            Dd = desc.dir[d]
            Dd.reverse = 1 if self.reverse[d] else 0
            Dd.xproj = c['xproj'].data_ptr(); Dd.wpack = c['wpkb'].data_ptr()
            for nm in ('gates', 'cst', 'hgates', 'hcst', 'aD'):
                setattr(Dd, nm, c[nm].data_ptr())
            dR = dY[:, d * N:(d + 1) * N]
            Dd.dr_ext = dR.data_ptr(); Dd.ld_dr = dR.stride(0)
            c['dXp'] = self._buf('dXp%d' % d, rows, 4 * N); Dd.dXp = c['dXp'].data_ptr()
            c['dXh'] = self._buf('dXh%d' % d, rows, 4 * H); Dd.dXh = c['dXh'].data_ptr()
            c['dA'] = self._buf('dA%d' % d, rows, 4 * N); Dd.dA = c['dA'].data_ptr()
            c['dD'] = self._buf('dD%d' % d, rows, 12 * N); Dd.dD = c['dD'].data_ptr()
            nsl = (B + 7) // 8                         # per batch chunk; unused slots stay zero
            c['dcv'] = self._buf('dcv%d' % d, nsl, 12 * N); c['dcv'].zero_(); Dd.dcv = c['dcv'].data_ptr()
            c['dbh'] = self._buf('dbh%d' % d, nsl, 4 * H); c['dbh'].zero_(); Dd.dbh = c['dbh'].data_ptr()
        net.scan_hyper(self, desc, backward=True)
        dX = self._buf('dX', rows, self.D) if need_dx else None
        if need_dx:                                   # critical path: feeds the scan of the layer below
            for d, dp in enumerate(self.dirs):
                c = ctx['d'][d]
                ops.gemm(c['dXp'], dp.p('Wx'), dX, tb=True, beta=0.0 if d == 0 else 1.0)
                ops.gemm(c['dXh'], dp.p('Wxh')[N:], dX, tb=True, beta=1.0)
        with net.deferred():
            self._param_grads(ctx, rows, T, B)
        return dX

    def _param_grads(self, ctx, rows, T, B):
        N, H = self.N, self.Hp
        X = ctx['X']
        n1 = (T - 1) * B
        for d, dp in enumerate(self.dirs):
            c = ctx['d'][d]
            R, Hh = c['R'], c['Hh']
            dA, dXh, dXp, dD = c['dA'], c['dXh'], c['dXp'], c['dD']
            if not self.reverse[d]:
                sl_prev, sl_cur = slice(0, n1), slice(B, rows)
            else:
                sl_prev, sl_cur = slice(B, rows), slice(0, n1)
            ops.gemm(R[sl_prev], dA[sl_cur], dp.g('Wh'), ta=True)
            ops.gemm(X, dXp, dp.g('Wx'), ta=True)
            gWxh = dp.g('Wxh')
            ops.gemm(R[sl_prev], dXh[sl_cur], gWxh[:N], ta=True)
            ops.gemm(X, dXh, gWxh[N:], ta=True)
            ops.colreduce(c['dbh'], dp.g('bh'), c['dbh'].shape[0], 4 * H)     # summed over time in the scan
            ops.gemm(Hh[sl_prev], dXh[sl_cur], dp.g('Whh'), ta=True)
            gM = dp.g('M')
            ops.gemm(Hh, dD, gM[:H], ta=True)
            ops.colreduce(c['dcv'], gM[H], c['dcv'].shape[0], 12 * N)

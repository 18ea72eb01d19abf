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
        if N % 4 or self.Hp > N:
            raise ValueError('asr_amd: ScalingHyperLSTMLayer needs outer units divisible by 4 and inner units (padded to a '
                             'multiple of 4: %d) not above the outer units (%d): a limit of the hyper scan' % (self.Hp, N))
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


# ------------------------------------------------------------------------------------------------
# The layer with its options (use_peepholes / use_layer_norm / learn_init): step-loop path
# ------------------------------------------------------------------------------------------------
_LN5 = G4 + ('outcell',)
_MAPS = ('in', 'hid', 'bias', 'cell')          # column-block order of S / s


def _mk_scaling_step(store, prefix, D, N, H, peep, ln, learn_init):
    """Engine tensors (gate-major, as asr_scaling_dir wants them) and the reference parameters as views of them, in
    the add_param order of specs.scaling_hyper_params.  Blocks the reference keeps as non-trainable zero constants
    (the cell gate's peephole and scale_outer_cell blocks) exist in the engine tensors, start at zero and receive
    zero gradients."""
    import ctypes  # noqa: F401  (kept local: this module is imported by builders that never take this path)
    NS = 4 if peep else 3
    dp = DirParams(store, prefix)
    dp.add('Wix', (D, 4 * H)); dp.add('Wio', (H + N, 4 * H)); dp.add('bi', (4 * H,))
    if peep:
        dp.add('wpeep', (4 * H,))
    if ln:
        dp.add('lna_i', (5, H)); dp.add('lnb_i', (5, H))
    if learn_init:
        dp.add('init', (2 * H + 2 * N,))
    dp.add('Wox', (D, 4 * N)); dp.add('Woo', (N, 4 * N)); dp.add('S', (H, NS * 4 * N)); dp.add('s', (NS * 4 * N,))
    if ln:
        dp.add('lna_o', (5, N)); dp.add('lnb_o', (5, N))
    cb = lambda k, U: (lambda E: E[:, k * U:(k + 1) * U])
    rcb = lambda r0, r1, k, U: (lambda E: E[r0:r1, k * U:(k + 1) * U])
    vb = lambda k, U: (lambda E: E[k * U:(k + 1) * U])
    row = lambda k: (lambda E: E[k])
    seg = lambda a, b: (lambda E: E[a:b].view(1, b - a))
    table = {}
    for g, gn in enumerate(G4):
        table['W_inner_in_to_inner_' + gn] = ('Wix', cb(g, H))
        table['W_inner_hid_to_inner_' + gn] = ('Wio', rcb(0, H, g, H))
        table['W_outer_hid_to_inner_' + gn] = ('Wio', rcb(H, H + N, g, H))
        table['W_inner_cell_to_inner_' + gn] = ('wpeep', vb(g, H))
        table['b_inner_' + gn] = ('bi', vb(g, H))
        table['W_outer_in_to_outer_' + gn] = ('Wox', cb(g, N))
        table['W_outer_hid_to_outer_' + gn] = ('Woo', cb(g, N))
        for m, mn in enumerate(_MAPS):
            table['W_inner_hid_to_outer_%s_%s' % (mn, gn)] = ('S', cb(4 * m + g, N))
            table['b_inner_hid_to_outer_%s_%s' % (mn, gn)] = ('s', vb(4 * m + g, N))
    for k, gn in enumerate(_LN5):
        table['W_inner_ln_' + gn] = ('lna_i', row(k)); table['b_inner_ln_' + gn] = ('lnb_i', row(k))
        table['W_outer_ln_' + gn] = ('lna_o', row(k)); table['b_outer_ln_' + gn] = ('lnb_o', row(k))
    table.update(inner_cell_init=('init', seg(0, H)), inner_hid_init=('init', seg(H, 2 * H)),
                 outer_cell_init=('init', seg(2 * H, 2 * H + N)), outer_hid_init=('init', seg(2 * H + N, 2 * H + 2 * N)))
    for name, _, _ in specs.scaling_hyper_params(D, H, N, peep, ln, learn_init):
        key, fn = table[name]
        dp.map(name, key, 'fn', fn)
    return dp


class ScalingStepPair(object):
    """ScalingHyperLSTMLayer pair with any of use_peepholes / use_layer_norm / learn_init, on asr_scaling_fwd/bwd
    (asr_amd/csrc/step_scaling.hip).  Host side as for ProjectionHyperLSTMLayer: the two all-timestep input
    projections before the loop, dX and every parameter gradient as GEMMs / column reductions over all T after it.
    Gate-major layout throughout (no interleave, no padding of the inner width beyond % 4)."""
    kind = 'scaling'
    stateful = False

    def __init__(self, net, idx, D, N, H, clip, ndir, use_peepholes, use_layer_norm, learn_init):
        if N % 4 or H % 4:
            raise ValueError('asr_amd: ScalingHyperLSTMLayer with options needs both unit counts divisible by 4')
        self.net = net; self.idx = idx
        self.D, self.N, self.H = D, N, H
        self.peep, self.ln, self.learn_init = bool(use_peepholes), bool(use_layer_norm), bool(learn_init)
        self.NS = 4 if self.peep else 3
        self.clip = float(clip); self.ndir = ndir
        self.out_dim = ndir * N
        self.dirs = [_mk_scaling_step(net.store, 'L%d.%d.' % (idx, d), D, N, H, self.peep, self.ln, self.learn_init)
                     for d in range(ndir)]
        self.kdirs = self.dirs               # what the kernels and the gradient products see (EqualScaleStepPair folds)
        self.reverse = [False, True][:ndir]
        self.inner_grad = [None] * ndir

    def describe(self):
        out = {}
        for d, nm in enumerate(('fwd', 'bwd')):
            if d >= self.ndir:
                out[nm] = None
                continue
            out[nm] = dict(kind='scaling',
                           args=dict(num_inner_units=self.H, num_outer_units=self.N, grad_clipping=self.clip,
                                     backwards=self.reverse[d], use_peepholes=self.peep, use_layer_norm=self.ln,
                                     learn_init=self.learn_init),
                           params=[r[0] for r in self.dirs[d].refmap])
        return out

    def _buf(self, name, rows, cols):
        return self.net.arena.mat('L%d.%s' % (self.idx, name), rows, cols)

    def inner_hid(self, d):
        return self.ctx['d'][d]['hid'][:, :self.H]

    def _fwd_bufs(self):
        N, H = self.N, self.H
        b = [('hid', H + N), ('gi', 4 * H), ('go', 4 * N), ('ci', H), ('co', N), ('sc', self.NS * 4 * N), ('a', 4 * N)]
        if self.ln:
            b += [('xh_i', 5 * H), ('xh_o', 5 * N), ('rs_i', 5), ('rs_o', 5)]
        return b

    def _bwd_bufs(self):
        N, H = self.N, self.H
        b = [('dh', H + N), ('dgi', 4 * H), ('dgo', 4 * N), ('dsc', self.NS * 4 * N), ('da', 4 * N)]
        if self.ln:
            b += [('dy_i', 5 * H), ('dy_o', 5 * N)]
        if self.peep:
            b += [('dpeep', 3 * H)]
        return b

    def _desc(self, T, B, mask, backward):
        desc = _lib.ScalingDesc()
        desc.T, desc.N, desc.H, desc.ndir, desc.Btot = T, self.N, self.H, self.ndir, B
        desc.use_peepholes, desc.use_layer_norm = int(self.peep), int(self.ln)
        desc.grad_clip = self.clip
        desc.mask = mask.data_ptr(); desc.mask_stride_b = mask.stride(0); desc.mask_stride_t = mask.stride(1)
        for d, dp in enumerate(self.kdirs):
            c = self.ctx['d'][d]
            Dd = desc.dir[d]
            Dd.reverse = 1 if self.reverse[d] else 0
            for f, key in (('Wio', 'Wio'), ('S', 'S'), ('s', 's'), ('Woo', 'Woo'), ('wpeep', 'wpeep'), ('lna_i', 'lna_i'),
                           ('lnb_i', 'lnb_i'), ('lna_o', 'lna_o'), ('lnb_o', 'lnb_o'), ('init', 'init')):
                if dp.has(key):
                    setattr(Dd, f, dp.p(key).data_ptr())
            for f in ('inner_in', 'outer_in') + tuple(n for n, _ in self._fwd_bufs()):
                setattr(Dd, f, c[f].data_ptr())
            if backward:
                for f in ('dhid',) + tuple(n for n, _ in self._bwd_bufs()):
                    setattr(Dd, f, c[f].data_ptr())
                if self.learn_init:
                    Dd.dinit = c['dinit'].data_ptr()
        return desc

    def _run(self, desc, B, backward):
        import ctypes as C
        L = _lib.lib()
        need = L.asr_scaling_ws_bytes(min(B, 64), self.N, self.H, self.ndir)
        ws = self.net._workspace(('scaling', self.idx), need)
        fn = L.asr_scaling_bwd if backward else L.asr_scaling_fwd
        for b0 in range(0, B, 64):                      # batch rows are independent: <= 64 rows per call
            desc.b0 = b0; desc.B = min(64, B - b0)
            check(fn(C.byref(desc), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                  'asr_scaling_%s' % ('bwd' if backward else 'fwd'))
        # the persistent launch's sync block heads the workspace: Network.check_status reads its abort word
        self.net._last_groups = getattr(self.net, '_last_groups', {})
        self.net._last_groups[('scaling', self.idx)] = 1

    def forward(self, X, T, B, mask, cond, train):
        N, H = self.N, self.H
        rows = T * B
        Y = self._buf('Y', rows, self.out_dim)
        self.ctx = ctx = dict(X=X, T=T, B=B, mask=mask, Y=Y, d=[])
        for d, dp in enumerate(self.kdirs):
            c = {}
            c['inner_in'] = ops.gemm(X, dp.p('Wix'), self._buf('iin%d' % d, rows, 4 * H), bias=dp.p('bi'))
            c['outer_in'] = ops.gemm(X, dp.p('Wox'), self._buf('oin%d' % d, rows, 4 * N))
            for nm, w in self._fwd_bufs():
                c[nm] = self._buf('%s%d' % (nm, d), rows, w)
            ctx['d'].append(c)
        self._run(self._desc(T, B, mask, False), B, False)
        for d in range(self.ndir):                      # SliceLayer(num_inner_units:) + ConcatLayer
            Y[:, d * N:(d + 1) * N].copy_(ctx['d'][d]['hid'][:, H:])
        return Y

    def backward(self, dY, need_dx=True):
        net = self.net; ctx = self.ctx; N, H = self.N, self.H
        X, T, B, mask = ctx['X'], ctx['T'], ctx['B'], ctx['mask']
        rows = T * B
        for d in range(self.ndir):
            c = ctx['d'][d]
            c['dhid'] = self._buf('dhid%d' % d, rows, H + N)
            if self.inner_grad[d] is None:
                c['dhid'][:, :H].zero_()
            else:
                c['dhid'][:, :H].copy_(self.inner_grad[d]); self.inner_grad[d] = None
            c['dhid'][:, H:].copy_(dY[:, d * N:(d + 1) * N])
            for nm, w in self._bwd_bufs():
                c[nm] = self._buf('%s%d' % (nm, d), rows, w)
            if self.learn_init:
                c['dinit'] = self._buf('dinit%d' % d, B, 2 * H + 2 * N)
        self._run(self._desc(T, B, mask, True), B, True)
        dX = self._buf('dX', rows, self.D) if need_dx else None
        if need_dx:
            for d, dp in enumerate(self.kdirs):
                c = ctx['d'][d]
                ops.gemm(c['dgi'], dp.p('Wix'), dX, tb=True, beta=0.0 if d == 0 else 1.0)
                ops.gemm(c['dgo'], dp.p('Wox'), dX, tb=True, beta=1.0)
        with net.deferred():
            self._param_grads(ctx, rows, T, B)
        return dX

    def _param_grads(self, ctx, rows, T, B):
        N, H = self.N, self.H
        X = ctx['X']
        n1 = (T - 1) * B
        for d, dp in enumerate(self.kdirs):
            c = ctx['d'][d]
            hid = c['hid']
            if not self.reverse[d]:
                sl_prev, sl_cur, sl_first = slice(0, n1), slice(B, rows), slice(0, B)
            else:
                sl_prev, sl_cur, sl_first = slice(B, rows), slice(0, n1), slice(n1, rows)
            ops.gemm(X, c['dgi'], dp.g('Wix'), ta=True)
            ops.colreduce(c['dgi'], dp.g('bi'), rows, 4 * H)
            ops.gemm(X, c['dgo'], dp.g('Wox'), ta=True)
            if n1:
                ops.gemm(hid[sl_prev], c['dgi'][sl_cur], dp.g('Wio'), ta=True)
                ops.gemm(hid[sl_prev][:, H:], c['da'][sl_cur], dp.g('Woo'), ta=True)
            else:                                       # T == 1: no recurrent step ran
                dp.g('Wio').zero_(); dp.g('Woo').zero_()
            if self.learn_init:                         # the first step's "previous hid" is the learned row, not zero
                ini = dp.p('init')
                h0 = self._buf('h0_%d' % d, B, H + N)
                h0[:, :H].copy_(ini[H:2 * H].expand(B, H)); h0[:, H:].copy_(ini[2 * H + N:].expand(B, N))
                ops.gemm(h0, c['dgi'][sl_first], dp.g('Wio'), ta=True, beta=1.0)
                ops.gemm(h0[:, H:], c['da'][sl_first], dp.g('Woo'), ta=True, beta=1.0)
                ops.colreduce(c['dinit'], dp.g('init'), B, 2 * H + 2 * N)
            ops.gemm(hid[:, :H], c['dsc'], dp.g('S'), ta=True)
            ops.colreduce(c['dsc'], dp.g('s'), rows, self.NS * 4 * N)
            if self.peep:
                gw = dp.g('wpeep')
                gw.zero_()                              # the cell gate has no peephole (cell_trainable=False)
                for k, blk in ((0, 0), (1, 1), (2, 3)):
                    ops.colreduce(c['dpeep'][:, k * H:(k + 1) * H], gw[blk * H:(blk + 1) * H], rows, H, ldx=3 * H)
            if self.ln:
                ops.colreduce(c['dy_i'], dp.g('lna_i').view(-1), rows, 5 * H, y=c['xh_i'])
                ops.colreduce(c['dy_i'], dp.g('lnb_i').view(-1), rows, 5 * H)
                ops.colreduce(c['dy_o'], dp.g('lna_o').view(-1), rows, 5 * N, y=c['xh_o'])
                ops.colreduce(c['dy_o'], dp.g('lnb_o').view(-1), rows, 5 * N)


# ------------------------------------------------------------------------------------------------
# EqualScaleHyperLSTMLayer (libs/lasagne_libs/hyper_layers.py:1207-1890)
# ------------------------------------------------------------------------------------------------
def _mk_equal_scale(store, prefix, D, N, H, peep, ln, learn_init=False):
    """Store tensors in the reference's shapes (gate-major concatenations) and the reference parameters as views of them,
    in the add_param order of specs.equal_scale_hyper_params.  U = N is the inner cell's width, H the projected one."""
    NS = 4 if peep else 3
    U = N
    dp = DirParams(store, prefix)
    dp.add('Wix', (D, 4 * U)); dp.add('Whh', (H, 4 * U)); dp.add('Woi', (N, 4 * U)); dp.add('bi', (4 * U,))
    if peep:
        dp.add('wpeep', (4 * U,))
    if ln:
        dp.add('lna_i', (5, U)); dp.add('lnb_i', (5, U))
    dp.add('P', (U, H))
    if learn_init:                 # the kernels' initial rows [inner cell | unprojected inner hid = 0 | outer cell | outer hid]
        dp.add('init', (2 * U + 2 * N,)); dp.add('h_init', (1, H))
    dp.add('Wox', (D, 4 * N)); dp.add('Woo', (N, 4 * N)); dp.add('S', (H, NS * 4 * N)); dp.add('s', (NS * 4 * N,))
    if ln:
        dp.add('lna_o', (5, N)); dp.add('lnb_o', (5, N))
    cb = lambda k, w: (lambda E: E[:, k * w:(k + 1) * w])
    vb = lambda k, w: (lambda E: E[k * w:(k + 1) * w])
    row = lambda k: (lambda E: E[k])
    table = {'W_inner_hid_prj': ('P', lambda E: E)}
    for g, gn in enumerate(G4):
        table['W_inner_in_to_inner_' + gn] = ('Wix', cb(g, U))
        table['W_inner_hid_to_inner_' + gn] = ('Whh', cb(g, U))
        table['W_outer_hid_to_inner_' + gn] = ('Woi', cb(g, U))
        table['W_inner_cell_to_inner_' + gn] = ('wpeep', vb(g, U))
        table['b_inner_' + gn] = ('bi', vb(g, U))
        table['W_outer_in_to_outer_' + gn] = ('Wox', cb(g, N))
        table['W_outer_hid_to_outer_' + gn] = ('Woo', cb(g, N))
        for m, mn in enumerate(_MAPS):
            table['W_inner_hid_to_outer_%s_%s' % (mn, gn)] = ('S', cb(4 * m + g, N))
            table['b_inner_hid_to_outer_%s_%s' % (mn, gn)] = ('s', vb(4 * m + g, N))
    for k, gn in enumerate(_LN5):
        table['W_inner_ln_' + gn] = ('lna_i', row(k)); table['b_inner_ln_' + gn] = ('lnb_i', row(k))
        table['W_outer_ln_' + gn] = ('lna_o', row(k)); table['b_outer_ln_' + gn] = ('lnb_o', row(k))
    seg = lambda a, b: (lambda E: E[a:b].view(1, -1))
    table.update(inner_cell_init=('init', seg(0, U)), inner_hid_init=('h_init', lambda E: E),
                 outer_cell_init=('init', seg(2 * U, 2 * U + N)), outer_hid_init=('init', seg(2 * U + N, 2 * U + 2 * N)))
    for name, _, _ in specs.equal_scale_hyper_params(D, H, N, peep, ln, learn_init):
        key, fn = table[name]
        dp.map(name, key, 'fn', fn)
    return dp


class _FoldedDir(object):
    """One direction as asr_scaling_* and ScalingStepPair's gradient products see it: `Wio` and `S` are the folded
    products (scratch, rebuilt every forward), their gradients scratch as well; everything else is the store's."""
    def __init__(self, dp):
        self.dp = dp
        self.fold = {}; self.gfold = {}

    def p(self, k):
        return self.fold[k] if k in self.fold else self.dp.p(k)

    def g(self, k):
        return self.gfold[k] if k in self.gfold else self.dp.g(k)

    def has(self, k):
        return k in ('Wio', 'S') or self.dp.has(k)


class EqualScaleStepPair(ScalingStepPair):
    """EqualScaleHyperLSTMLayer pair (hyper_layers.py:1207-1890) on the ScalingHyperLSTMLayer step loop
    (asr_scaling_fwd/bwd).  The layer is ScalingHyperLSTMLayer's step with an inner cell of U = num_outer_units columns
    per gate whose output u_t = sigmoid(outgate)*tanh(outcell) is projected, h_in = u_t.P (P = W_inner_hid_prj, U x H),
    before it is carried (:1709).  h_in only ever enters products, and the masked carry is linear in it, so the projection
    folds into its consumers:  h_in.W_inner_hid_to_inner = u.(P.W_hh)  (:1657),  h_in.[scale maps] = u.(P.S)  (:1714-1717).
    The kernels therefore run a ScalingHyperLSTMLayer of inner width U with the folded weights; the inner hidden sequence
    is one GEMM after the loop, and the factors' gradients come from the folded ones by the product rule:
      dW_hh = P^T.dA,  dS = P^T.dS',  dP = dA.W_hh^T + dS'.S^T.
    learn_init (:1782-1788): inner_hid_init is a row of the PROJECTED space, so it has no place in the unprojected state the
    kernels carry (that segment of their initial rows stays zero).  It reaches the recurrence only as
    inner_hid_init.W_inner_hid_to_inner at the first unmasked step of a row: the kernels add that row (asr_scaling_dir.tg0)
    while a row's previous state is still the initial one, and hand back the inner-gate gradients of those steps (dgi0),
    from which d inner_hid_init = (sum_b dgi0).W_hh^T and dW_hh += inner_hid_init^T.(sum_b dgi0).  The projected inner
    hidden sequence holds inner_hid_init itself on the reverse direction's padding (masked steps before a row's first)."""
    kind = 'equalscale'

    def __init__(self, net, idx, D, N, H, clip, ndir, use_peepholes, use_layer_norm, learn_init):
        if N % 4 or H % 4:
            raise ValueError('asr_amd: EqualScaleHyperLSTMLayer needs both unit counts divisible by 4')
        self.net = net; self.idx = idx
        self.D, self.N, self.Href = D, N, H
        self.H = N                                  # the kernels' inner width
        self.peep, self.ln, self.learn_init = bool(use_peepholes), bool(use_layer_norm), bool(learn_init)
        self.NS = 4 if self.peep else 3
        self.clip = float(clip); self.ndir = ndir
        self.out_dim = ndir * N
        self.dirs = [_mk_equal_scale(net.store, 'L%d.%d.' % (idx, d), D, N, H, self.peep, self.ln, self.learn_init)
                     for d in range(ndir)]
        self.kdirs = [_FoldedDir(dp) for dp in self.dirs]
        self.reverse = [False, True][:ndir]
        self.inner_grad = [None] * ndir

    def describe(self):
        out = ScalingStepPair.describe(self)
        for d, nm in enumerate(('fwd', 'bwd')):
            if out[nm] is not None:
                out[nm]['kind'] = 'equalscale'
                out[nm]['args']['num_inner_units'] = self.Href
        return out

    def inner_hid(self, d):
        """(T*B, num_inner_units): the projected inner hidden sequence of the last forward."""
        u = self.ctx['d'][d]['hid'][:, :self.H]
        ih = ops.gemm(u, self.dirs[d].p('P'), self._buf('ihid%d' % d, u.shape[0], self.Href))
        if self.learn_init and self.reverse[d]:        # padding of the reverse scan: the carried state is inner_hid_init
            B, Hr = self.ctx['B'], self.Href
            one = self._buf('one_bh', B, Hr); one.fill_(1.0)
            ones = self._buf('ones_col', B, 1); ones.fill_(1.0)
            h0 = ops.gemm(ones, self.dirs[d].p('h_init'), self._buf('h0b%d' % d, B, Hr))
            nh0 = ops.gemm(ones, self.dirs[d].p('h_init'), self._buf('nh0b%d' % d, B, Hr), alpha=-1.0)
            ops.affine_rows(ih, Hr, one, h0, nh0, self.ctx['mask'], B, ih, Hr, ih.shape[0], Hr)
        return ih

    def forward(self, X, T, B, mask, cond, train):
        U, Hr, N = self.H, self.Href, self.N
        for d, (dp, k) in enumerate(zip(self.dirs, self.kdirs)):
            Wio = self._buf('Wio_f%d' % d, U + N, 4 * U)
            ops.gemm(dp.p('P'), dp.p('Whh'), Wio[:U])
            Wio[U:].copy_(dp.p('Woi'))
            k.fold['Wio'] = Wio
            k.fold['S'] = ops.gemm(dp.p('P'), dp.p('S'), self._buf('S_f%d' % d, U, self.NS * 4 * N))
            k.gfold['Wio'] = self._buf('gWio_f%d' % d, U + N, 4 * U)
            k.gfold['S'] = self._buf('gS_f%d' % d, U, self.NS * 4 * N)
            if self.learn_init:
                k.tg0 = ops.gemm(dp.p('h_init'), dp.p('Whh'), self._buf('tg0_%d' % d, 1, 4 * U))
        return ScalingStepPair.forward(self, X, T, B, mask, cond, train)

    def _desc(self, T, B, mask, backward):
        desc = ScalingStepPair._desc(self, T, B, mask, backward)
        if self.learn_init:
            for d, k in enumerate(self.kdirs):
                desc.dir[d].tg0 = k.tg0.data_ptr()
                if backward:
                    k.dgi0 = self._buf('dgi0_%d' % d, B, 4 * self.H); k.dgi0.zero_()
                    desc.dir[d].dgi0 = k.dgi0.data_ptr()
        return desc

    def backward(self, dY, need_dx=True):
        if any(g is not None for g in self.inner_grad):     # no caller differentiates through the inner hidden sequences
            raise NotImplementedError('asr_amd: a gradient on EqualScaleHyperLSTMLayer\'s inner hidden sequences is not built')
        return ScalingStepPair.backward(self, dY, need_dx)

    def _param_grads(self, ctx, rows, T, B):
        ScalingStepPair._param_grads(self, ctx, rows, T, B)
        U = self.H
        for d, (dp, k) in enumerate(zip(self.dirs, self.kdirs)):
            gW, gS, P = k.gfold['Wio'], k.gfold['S'], dp.p('P')
            ops.gemm(P, gW[:U], dp.g('Whh'), ta=True)
            dp.g('Woi').copy_(gW[U:])
            ops.gemm(P, gS, dp.g('S'), ta=True)
            ops.gemm(gW[:U], dp.p('Whh'), dp.g('P'), tb=True)
            ops.gemm(gS, dp.p('S'), dp.g('P'), tb=True, beta=1.0)
            if self.learn_init:
                dp.g('init')[U:2 * U].zero_()           # the unprojected inner hid rows are no parameter
                srow = ops.colreduce(k.dgi0, self._buf('sdgi0_%d' % d, 1, 4 * U).view(-1), B, 4 * U).view(1, 4 * U)
                ops.gemm(dp.p('h_init'), srow, dp.g('Whh'), ta=True, beta=1.0)
                ops.gemm(srow, dp.p('Whh'), dp.g('h_init'), tb=True)

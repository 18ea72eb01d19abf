"""HyperLSTM layer pair (libs/lasagne_libs/hyper_lhuc_layers.py:168-507) on the persistent
HyperLSTM scan kernel.  See asr_amd/csrc/scan_hyper.hip for what is hoisted / folded."""
import ctypes as C
import torch

from . import _lib, ops, specs
from ._lib import check
from .engine import DirParams

G4 = specs.GATES_LHUC


def _mk_hyper(store, prefix, D, N, H, Z):
    dp = DirParams(store, prefix)
    dp.add('Whh', (H, 4 * H)); dp.add('Wxh', (N + D, 4 * H)); dp.add('bh', (4 * H,))
    dp.add('Wz_h', (4, H + 1, Z)); dp.add('Wz_x', (4, H + 1, Z)); dp.add('Wz_b', (4, H, Z))
    dp.add('Wzh', (4, Z, N)); dp.add('Wzx', (4, Z, N)); dp.add('Wzb', (4, Z, N)); dp.add('b0', (4, N))
    dp.add('Wh', (N, 4 * N)); dp.add('Wx', (D, 4 * N))
    # reference add_param order: hyper_lhuc_layers.py:251-276
    for g, gn in enumerate(G4):
        dp.map('W_hhat_' + gn, 'Whh', 'gatecols', g); dp.map('W_xhat_' + gn, 'Wxh', 'gatecols', g)
        dp.map('bhat_' + gn, 'bh', 'gatevec', g)
    for g, gn in enumerate(G4):
        dp.map('W_hhat_h_' + gn, 'Wz_h', 'aug_w', g); dp.map('b_hhat_h_' + gn, 'Wz_h', 'aug_b', g)
        dp.map('W_hhat_x_' + gn, 'Wz_x', 'aug_w', g); dp.map('b_hhat_x_' + gn, 'Wz_x', 'aug_b', g)
        dp.map('W_hhat_b_' + gn, 'Wz_b', 'index', g)
    for g, gn in enumerate(G4):
        dp.map('W_hz_' + gn, 'Wzh', 'index', g); dp.map('W_xz_' + gn, 'Wzx', 'index', g)
        dp.map('W_bz_' + gn, 'Wzb', 'index', g); dp.map('b_0_' + gn, 'b0', 'index', g)
    for g, gn in enumerate(G4):
        dp.map('W_h_' + gn, 'Wh', 'gatecols', g); dp.map('W_x_' + gn, 'Wx', 'gatecols', g)
    return dp


def perm4(src, dst, rows, N, inverse=False):
    check(_lib.lib().asr_perm4(src.data_ptr(), dst.data_ptr(), rows, N, 1 if inverse else 0,
                               torch.cuda.current_stream().cuda_stream), 'asr_perm4')
    return dst


def chunk_bounds(T, n):
    """Time-chunk boundaries 0 = b_0 < ... < b_n = T, symmetric (b_i + b_{n-i} = T): the launch that runs
    steps [b_k, b_k+1) covers time chunk k of the forward direction and time chunk n-1-k of the reverse
    direction, both on the same grid."""
    n = max(1, min(int(n), T))
    b = [0] * (n + 1)
    for i in range(n + 1):
        b[i] = (i * T) // n if 2 * i <= n else T - ((n - i) * T) // n
    return [v for k, v in enumerate(b) if k == 0 or v > b[k - 1]]


class HyperPair(object):
    """Runs as `net.scan_chunks` consecutive launches over time chunks when training at T >= 256 (steps
    [step0, step1) of asr_hyper_scan_desc): the half of the NEXT layer's input projection (or of the output
    layer) that a finished chunk already determines runs on the side stream under the following chunks, and
    backward the same holds for dX; only the last chunk's share of those GEMMs stays on the critical path."""
    kind = 'hyper'
    stateful = False
    lhuc = 0                     # asr_hyper_scan_desc.lhuc_reparam
    _mk = staticmethod(_mk_hyper)
    input_ready = False
    chunked_forward = True

    def __init__(self, net, idx, D, N, H, Z, clip, ndir):
        self.net = net; self.idx = idx
        self.D, self.N, self.H, self.Z = D, N, H, Z
        self.clip = float(clip); self.ndir = ndir
        self.out_dim = ndir * N
        self.dirs = [self._mk(net.store, 'L%d.%d.' % (idx, d), D, N, H, Z) for d in range(ndir)]
        self.reverse = [False, True][:ndir]

    def _xbias(self, dp):
        return None

    def describe(self):
        out = {}
        for d, nm in enumerate(('fwd', 'bwd')):
            if d >= self.ndir:
                out[nm] = None
                continue
            out[nm] = dict(kind='hyper',
                           args=dict(num_units=self.N, num_hyper_units=self.H, num_proj_units=self.Z,
                                     grad_clipping=self.clip, backwards=self.reverse[d]),
                           params=[r[0] for r in self.dirs[d].refmap])
        return out

    def _buf(self, name, rows, cols):
        return self.net.arena.mat('L%d.%s' % (self.idx, name), rows, cols)

    def _fold(self, dp, d):
        """M (H+1, 12N) interleaved = [W_hhat_t ; b_t] . W_tz per gate  (eq.11 into eq.12)."""
        N, H = self.N, self.H
        Mg = self._buf('Mg%d' % d, H + 1, 12 * N).view(H + 1, 3, 4, N)
        for tau, (wz, wzz) in enumerate((('Wz_h', 'Wzh'), ('Wz_x', 'Wzx'))):
            for g in range(4):
                ops.gemm(dp.p(wz)[g], dp.p(wzz)[g], Mg[:, tau, g, :])
        for g in range(4):
            ops.gemm(dp.p('Wz_b')[g], dp.p('Wzb')[g], Mg[:H, 2, g, :])
        Mg[H, 2].copy_(dp.p('b0'))
        Mi = self._buf('Mi%d' % d, H + 1, 12 * N)
        perm4(Mg, Mi, (H + 1) * 3, N)
        return Mi

    # ---- consumer side: the input projection arrives in pieces from the layer below
    def begin_input(self, T, B):
        rows = T * B
        self._in = [(self._buf('xproj%d' % d, rows, 4 * self.N), self._buf('xhproj%d' % d, rows, 4 * self.H))
                    for d in range(self.ndir)]
        self.input_ready = True

    def accumulate_input(self, Xpart, r0, r1, k0, first):
        """rows [r0, r1) of xproj / xhproj (+)= Xpart . W[k0 : k0 + Xpart.shape[1]]  (bias with the first piece)"""
        N = self.N
        kw = Xpart.shape[1]
        for d, dp in enumerate(self.dirs):
            xp, xh = self._in[d]
            ops.gemm(Xpart, dp.p('Wx')[k0:k0 + kw], xp[r0:r1], beta=0.0 if first else 1.0,
                     bias=self._xbias(dp) if first else None)
            ops.gemm(Xpart, dp.p('Wxh')[N + k0:N + k0 + kw], xh[r0:r1], beta=0.0 if first else 1.0,
                     bias=dp.p('bh') if first else None)

    def _bounds(self, T, train):
        n = self.net.scan_chunks if (train and T >= 256) else 1
        return chunk_bounds(T, n)

    def prepare(self, train, T, B, mask, cond):
        """Weight-only forward work (engine.Network.prepare_ahead): the folded matrix M and the packed recurrent
        slices of both passes, per direction."""
        net = self.net; N, H = self.N, self.H
        L = _lib.lib()
        out = []
        for d, dp in enumerate(self.dirs):
            c = {}
            Mi = self._fold(dp, d)
            c['Mi'] = Mi
            for mode, nm in ((0, 'wpk'), (1, 'wpkb')):
                if mode == 1 and not train:
                    continue
                n = L.asr_hyper_pack_floats(N, H, mode)
                o = net.arena.get('L%d.%s%d' % (self.idx, nm, d), n)
                check(L.asr_hyper_pack_w(dp.p('Wh').data_ptr(), dp.p('Wxh').data_ptr(), dp.p('Whh').data_ptr(),
                                         Mi.data_ptr(), N, H, mode, o.data_ptr(),
                                         torch.cuda.current_stream().cuda_stream), 'asr_hyper_pack_w')
                c[nm] = o
            out.append(c)
        return out

    def forward(self, X, T, B, mask, cond, train, consumer=None):
        net = self.net; N, H = self.N, self.H
        rows = T * B
        L = _lib.lib()
        Y = self._buf('Y', rows, self.out_dim)
        desc = _lib.HyperScanDesc()
        desc.T, desc.B, desc.N, desc.H, desc.ndir = T, B, N, H, self.ndir
        desc.grad_clip = self.clip; desc.lhuc_reparam = self.lhuc
        desc.mask = mask.data_ptr(); desc.mask_stride_b = mask.stride(0); desc.mask_stride_t = mask.stride(1)
        self.ctx = ctx = dict(X=X, T=T, B=B, mask=mask, Y=Y, d=[], train=train)
        fed = self.input_ready
        self.input_ready = False
        prep = net.take_prepared(self) or self.prepare(train, T, B, mask, cond)
        for d, dp in enumerate(self.dirs):
            c = prep[d]
            Mi = c['Mi']
            if fed:                              # projected piecewise by the layer below (accumulate_input)
                c['xproj'], c['xhproj'] = self._in[d]
            else:
                c['xproj'] = ops.gemm(X, dp.p('Wx'), self._buf('xproj%d' % d, rows, 4 * N), bias=self._xbias(dp))
                c['xhproj'] = ops.gemm(X, dp.p('Wxh')[N:], self._buf('xhproj%d' % d, rows, 4 * H), bias=dp.p('bh'))
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
        bnd = self._bounds(T, train)
        n = len(bnd) - 1
        pipe = consumer is not None and n > 1
        if pipe:
            consumer.begin_input(T, B)
        for k in range(n):
            desc.step0, desc.step1 = bnd[k], bnd[k + 1]
            net.scan_hyper(self, desc, backward=False)
            if pipe:
                with net.deferred():
                    # forward direction finished time chunk k, reverse direction time chunk n-1-k; a time chunk's
                    # first piece is the one of the earlier launch (same launch: the forward direction's)
                    consumer.accumulate_input(Y[bnd[k] * B:bnd[k + 1] * B, 0:N], bnd[k] * B, bnd[k + 1] * B, 0,
                                              first=(self.ndir == 1 or 2 * k <= n - 1))
                    if self.ndir == 2:
                        j = n - 1 - k
                        consumer.accumulate_input(Y[bnd[j] * B:bnd[j + 1] * B, N:2 * N], bnd[j] * B, bnd[j + 1] * B, N,
                                                  first=(2 * k < n - 1))
        if pipe:
            net.join_deferred()
        return Y

    def backward(self, dY, need_dx=True):
        net = self.net; ctx = self.ctx; N, H, Z = self.N, self.H, self.Z
        X, T, B, mask = ctx['X'], ctx['T'], ctx['B'], ctx['mask']
        rows = T * B
        desc = _lib.HyperScanDesc()
        desc.T, desc.B, desc.N, desc.H, desc.ndir = T, B, N, H, self.ndir
        desc.grad_clip = self.clip; desc.lhuc_reparam = self.lhuc
        desc.mask = mask.data_ptr(); desc.mask_stride_b = mask.stride(0); desc.mask_stride_t = mask.stride(1)
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
            Dd.carry = self._buf('carry%d' % d, B, 2 * N + 2 * H).data_ptr()
        bnd = self._bounds(T, ctx.get('train', True))
        n = len(bnd) - 1
        dX = self._buf('dX', rows, self.D) if need_dx else None

        def dx_piece(d, j, first):
            """dX rows of time chunk j (+)= this direction's share: critical path (feeds the scan below)"""
            r0, r1 = bnd[j] * B, bnd[j + 1] * B
            dp = self.dirs[d]; c = ctx['d'][d]
            ops.gemm(c['dXp'][r0:r1], dp.p('Wx'), dX[r0:r1], tb=True, beta=0.0 if first else 1.0)
            ops.gemm(c['dXh'][r0:r1], dp.p('Wxh')[N:], dX[r0:r1], tb=True, beta=1.0)

        for k in range(n):
            desc.step0, desc.step1 = bnd[k], bnd[k + 1]
            net.scan_hyper(self, desc, backward=True)
            if need_dx and n > 1:
                # BPTT steps [b_k, b_k+1): the reverse direction has finished time chunk k, the forward direction
                # time chunk n-1-k; their dX shares run on the side stream under the next chunks
                with net.deferred():
                    if self.ndir == 2:
                        dx_piece(1, k, first=(2 * k <= n - 1))
                        dx_piece(0, n - 1 - k, first=(2 * k < n - 1))
                    else:
                        dx_piece(0, n - 1 - k, first=True)
        if need_dx and n > 1:
            net.join_deferred()
        elif need_dx:
            for d, dp in enumerate(self.dirs):
                c = ctx['d'][d]
                ops.gemm(c['dXp'], dp.p('Wx'), dX, tb=True, beta=0.0 if d == 0 else 1.0)
                ops.gemm(c['dXh'], dp.p('Wxh')[N:], dX, tb=True, beta=1.0)
        # parameter gradients: nothing below depends on them, so they run on the side stream under
        # the (latency-bound) scan of the layer below
        with net.deferred():
            self._param_grads(ctx, rows, T, B)
        return dX

    def _param_grads(self, ctx, rows, T, B):
        N, H = self.N, self.H
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
            ops.gemm(R[sl_prev], dA[sl_cur], dp.g('Wh'), ta=True)             # dW_h
            ops.gemm(X, dXp, dp.g('Wx'), ta=True)                             # dW_x
            gWxh = dp.g('Wxh')
            ops.gemm(R[sl_prev], dXh[sl_cur], gWxh[:N], ta=True)              # dW_xhat (h rows)
            ops.gemm(X, dXh, gWxh[N:], ta=True)                               # dW_xhat (x rows)
            ops.colreduce(c['dbh'], dp.g('bh'), c['dbh'].shape[0], 4 * H)     # summed over time in the scan
            ops.gemm(Hh[sl_prev], dXh[sl_cur], dp.g('Whh'), ta=True)          # dW_hhat
            self._fold_grads(dp, d, c, Hh, dD, rows)

    def _fold_grads(self, dp, d, c, Hh, dD, rows):
            N, H = self.N, self.H
            dMi = self._buf('dMi%d' % d, H + 1, 12 * N)
            ops.gemm(Hh, dD, dMi[:H], ta=True)
            ops.colreduce(c['dcv'], dMi[H], c['dcv'].shape[0], 12 * N)
            dMg = self._buf('dMg%d' % d, H + 1, 12 * N)
            perm4(dMi, dMg, (H + 1) * 3, N, inverse=True)
            dMg = dMg.view(H + 1, 3, 4, N)
            for tau, (wz, wzz) in enumerate((('Wz_h', 'Wzh'), ('Wz_x', 'Wzx'))):
                for g in range(4):
                    ops.gemm(dMg[:, tau, g, :], dp.p(wzz)[g], dp.g(wz)[g], tb=True)
                    ops.gemm(dp.p(wz)[g], dMg[:, tau, g, :], dp.g(wzz)[g], ta=True)
            for g in range(4):
                ops.gemm(dMg[:H, 2, g, :], dp.p('Wzb')[g], dp.g('Wz_b')[g], tb=True)
                ops.gemm(dp.p('Wz_b')[g], dMg[:H, 2, g, :], dp.g('Wzb')[g], ta=True)
            dp.g('b0').copy_(dMg[H, 2])


# ------------------------------------------------------------------------------------------------
REPARAM = {'relu': 1, 'exp': 2, 'identity': 3, '2sigmoid': 4}


def _mk_hyperlhuc(store, prefix, D, N, H, Z):
    dp = DirParams(store, prefix)
    dp.add('Whh', (H, 4 * H)); dp.add('Wxh', (N + D, 4 * H)); dp.add('bh', (4 * H,))
    dp.add('Wz', (H + 1, Z)); dp.add('Wzz', (Z, N))
    dp.add('Wh', (N, 4 * N)); dp.add('Wx', (D, 4 * N)); dp.add('b', (4 * N,))
    # reference add_param order: hyper_lhuc_layers.py:533-555, :521-525
    for g, gn in enumerate(G4):
        dp.map('W_hhat_' + gn, 'Whh', 'gatecols', g); dp.map('W_xhat_' + gn, 'Wxh', 'gatecols', g)
        dp.map('bhat_' + gn, 'bh', 'gatevec', g)
    dp.map('W_hhat_h', 'Wz', 'fn', lambda E: E[:-1]); dp.map('b_hhat_h', 'Wz', 'fn', lambda E: E[-1])
    dp.map('W_hz', 'Wzz')
    for g, gn in enumerate(G4):
        dp.map('W_h_' + gn, 'Wh', 'gatecols', g); dp.map('W_x_' + gn, 'Wx', 'gatecols', g)
        dp.map('b_' + gn, 'b', 'gatevec', g)
    return dp


class HyperLHUCPair(HyperPair):
    """HyperLHUCLSTMLayer (libs/lasagne_libs/hyper_lhuc_layers.py:509-678) on the HyperLSTM scan kernel
    in its lhuc mode: M (H+1, 12N) holds constants d_h = d_x = 1 (zero matrices, cvec rows = 1) and, in
    gate column 0 of the bias tile, the folded LHUC map [W_hhat_h ; b_hhat_h] . W_hz (H+1, N)."""
    kind = 'hyperlhuc'
    _mk = staticmethod(_mk_hyperlhuc)

    def __init__(self, net, idx, D, N, H, Z, clip, ndir, reparam='relu'):
        HyperPair.__init__(self, net, idx, D, N, H, Z, clip, ndir)
        self.reparam = reparam
        self.lhuc = REPARAM[reparam]
        self._tmpl = None

    def describe(self):
        out = HyperPair.describe(self)
        for nm in ('fwd', 'bwd'):
            if out[nm] is not None:
                out[nm]['kind'] = 'hyperlhuc'
                out[nm]['args']['reparam'] = self.reparam
        return out

    def _xbias(self, dp):
        return dp.p('b')

    def _fold(self, dp, d):
        N, H = self.N, self.H
        if self._tmpl is None:                   # constant part of M, gate-major: d_h = d_x = 1
            import numpy as np
            t = np.zeros((H + 1, 3, 4, N), np.float32)
            t[H, 0] = 1.0; t[H, 1] = 1.0
            self._tmpl = torch.from_numpy(t).to(self.net.device).view(H + 1, 12 * N)
        Mg = self._buf('Mg%d' % d, H + 1, 12 * N)
        Mg.copy_(self._tmpl)
        ops.gemm(dp.p('Wz'), dp.p('Wzz'), Mg.view(H + 1, 3, 4, N)[:, 2, 0, :])
        Mi = self._buf('Mi%d' % d, H + 1, 12 * N)
        perm4(Mg, Mi, (H + 1) * 3, N)
        return Mi

    def _fold_grads(self, dp, d, c, Hh, dD, rows):
        N, H = self.N, self.H
        dMi = self._buf('dMi%d' % d, H + 1, 12 * N)
        ops.gemm(Hh, dD, dMi[:H], ta=True)
        ops.colreduce(c['dcv'], dMi[H], c['dcv'].shape[0], 12 * N)
        dMg = self._buf('dMg%d' % d, H + 1, 12 * N)
        perm4(dMi, dMg, (H + 1) * 3, N, inverse=True)
        dMl = dMg.view(H + 1, 3, 4, N)[:, 2, 0, :]                      # d of the folded LHUC map
        ops.gemm(dMl, dp.p('Wzz'), dp.g('Wz'), tb=True)
        ops.gemm(dp.p('Wz'), dMl, dp.g('Wzz'), ta=True)
        ops.colreduce(c['dXp'], dp.g('b'), rows, 4 * N)                 # d b_stacked (:645)

"""ProjectionHyperLSTMLayer pair (libs/lasagne_libs/hyper_layers.py:14-523) on the step-loop HIP path of
asr_amd/csrc/step_prjhyper.hip.  Host side: the two all-timestep input projections (:316-317) and the folded
recurrent matrices (both projections multiplied into their consumers, see the kernel file) before the loop,
hid = R . W_prj, dX and every parameter gradient as large GEMMs over all T after it (deferred to the side
stream where nothing below depends on them); the gradients of the folded matrices are un-folded by the product
rule (exact algebra, checked against the oracle).  The layer's output fed upward is the OUTER hidden sequence of each
direction (models/gating_hyper_nets.py:150-171); the inner one is kept for `get_inner_hid`.
Gate vectors stay in the reference's gate-major layout here (no interleave)."""
import ctypes as C
import numpy as np
import torch

from . import _lib, ops, specs
from ._lib import check
from .engine import DirParams

G4 = specs.GATES_REC


def _mk_prj(store, prefix, D, N, F):
    dp = DirParams(store, prefix)
    dp.add('Wix', (D, 4 * N)); dp.add('Wio', (2 * F, 4 * N)); dp.add('bi', (4 * N,))
    dp.add('Wip', (N, F)); dp.add('ai', (N,)); dp.add('bic', (N,))
    dp.add('Wox', (D, 4 * N)); dp.add('Woo', (F, 4 * N)); dp.add('S', (F, 12 * N)); dp.add('s', (12 * N,))
    dp.add('Wop', (N, F)); dp.add('ao', (N,)); dp.add('boc', (N,))
    cb = lambda k: (lambda E: E[:, k * N:(k + 1) * N])            # column block k of a (., m*N) matrix
    rcb = lambda r0, r1, k: (lambda E: E[r0:r1, k * N:(k + 1) * N])
    vb = lambda k: (lambda E: E[k * N:(k + 1) * N])
    # reference add_param order: hyper_layers.py:55-113 (inner), :118-208 (outer)
    for g, gn in enumerate(G4):
        dp.map('W_in_to_inner_' + gn, 'Wix', 'fn', cb(g))
        dp.map('W_inner_to_inner_' + gn, 'Wio', 'fn', rcb(0, F, g))
        dp.map('W_outer_to_inner_' + gn, 'Wio', 'fn', rcb(F, 2 * F, g))
        dp.map('b_inner_' + gn, 'bi', 'fn', vb(g))
    dp.map('W_inner_prj', 'Wip'); dp.map('W_inner_cell_ln', 'ai'); dp.map('b_inner_cell_ln', 'bic')
    for g, gn in enumerate(G4):
        dp.map('W_in_to_outer_' + gn, 'Wox', 'fn', cb(g))
        dp.map('W_outer_to_outer_' + gn, 'Woo', 'fn', cb(g))
        for k, nm in enumerate(('in', 'outer', 'bias')):
            dp.map('W_inner_to_%s_%s' % (nm, gn), 'S', 'fn', cb(4 * k + g))
            dp.map('b_inner_to_%s_%s' % (nm, gn), 's', 'fn', vb(4 * k + g))
    dp.map('W_outer_prj', 'Wop'); dp.map('W_outer_cell_ln', 'ao'); dp.map('b_outer_cell_ln', 'boc')
    return dp


class PrjHyperPair(object):
    kind = 'prjhyper'
    stateful = False

    def __init__(self, net, idx, D, N, F, clip, ndir):
        if N % 4 or F % 4:
            raise ValueError('asr_amd: ProjectionHyperLSTMLayer needs num_units and num_factors divisible by 4')
        self.net = net; self.idx = idx
        self.D, self.N, self.F = D, N, F
        self.clip = float(clip); self.ndir = ndir
        self.out_dim = ndir * F
        self.dirs = [_mk_prj(net.store, 'L%d.%d.' % (idx, d), D, N, F) for d in range(ndir)]
        self.reverse = [False, True][:ndir]
        self.inner_grad = [None] * ndir   # gradient on the inner hidden sequences, consumed by the next backward

    def describe(self):
        out = {}
        for d, nm in enumerate(('fwd', 'bwd')):
            if d >= self.ndir:
                out[nm] = None
                continue
            out[nm] = dict(kind='prjhyper',
                           args=dict(num_factors=self.F, num_units=self.N, grad_clipping=self.clip,
                                     backwards=self.reverse[d], only_outer=True),
                           params=[r[0] for r in self.dirs[d].refmap])
        return out

    def _buf(self, name, rows, cols):
        return self.net.arena.mat('L%d.%s' % (self.idx, name), rows, cols)

    def inner_hid(self, d):
        return self.ctx['d'][d]['hid'][:, :self.F]

    def aux_costs(self, w_inter=0.0, w_intra=0.0):
        """scripts/hypernets/exp_prj_hyper.py:108-124 on the inner hidden sequences of the last forward:
        returns a device tensor (train_inter_cost, train_intra_cost); with a non-zero weight the gradient of
        w_inter*inter + w_intra*intra with respect to those sequences is left for the next backward.
        Everything on the device: per-row statistics (masked mean: asr_seq_pool_fwd mode 2; unmasked sum and sum of
        squares over time: asr_colreduce with gmod = B), the (B,F) algebra and both costs by asr_var_costs, the
        gradient by asr_affine_rows.  Data parallel: the costs are statistics of the GLOBAL batch, so the ranks
        all-gather their (3,B,F) statistics, lengths and padded T (each rank keeps the padding of its own shard) and
        every rank forms the global costs and the coefficients of its own rows."""
        ctx = self.ctx; T, B, mask = ctx['T'], ctx['B'], ctx['mask']
        F = self.F; rows = T * B
        world = getattr(self.net, 'world', 1)
        rank = getattr(self.net, 'dropout_rank', 0) if world > 1 else 0
        want = (w_inter != 0.0) or (w_intra != 0.0)
        st = self._buf('aux_st', 3 * self.ndir * B, F)
        lens = self._buf('aux_len', 1, B)
        torch.sum(mask, dim=1, out=lens[0])                       # plumbing: the utterance lengths
        costs = self._buf('aux_costs', 1, 2)
        for d in range(self.ndir):
            feat = self.inner_hid(d)                                  # (rows, F) view, leading dimension 2F
            o = 3 * d * B
            ops.seq_pool_fwd(feat, 2 * F, B * 2 * F, mask.contiguous(), B, T, F, 2, st[o:o + B])
            ops.colreduce(feat, st[o + B:o + 2 * B], rows, F, gmod=B, ldx=2 * F)
            ops.colreduce(feat, st[o + 2 * B:o + 3 * B], rows, F, y=feat, gmod=B, ldx=2 * F, ldy=2 * F)
        Bg, b0, stg, leng, tpad = B, 0, st, lens[0], None
        if world > 1:
            import torch.distributed as dist
            Bg, b0 = B * world, B * rank
            parts = [torch.empty_like(st) for _ in range(world)]
            dist.all_gather(parts, st)                                # (world) x (ndir, 3, B, F)
            stg = torch.stack([p_.view(self.ndir, 3, B, F) for p_ in parts], 2).reshape(self.ndir * 3 * Bg, F).contiguous()
            meta = torch.cat([lens[0], torch.full((B,), float(T), device=st.device)])
            metas = [torch.empty_like(meta) for _ in range(world)]
            dist.all_gather(metas, meta)
            leng = torch.cat([m_[:B] for m_ in metas]).contiguous()
            tpad = torch.cat([m_[B:] for m_ in metas]).contiguous()
        for d in range(self.ndir):
            coef = self._buf('aux_coef%d' % d, 3 * B, F) if want else None
            ops.var_costs(stg[3 * d * Bg:3 * (d + 1) * Bg], leng, tpad, Bg, F, T, float(w_inter), float(w_intra), b0, B,
                          coef, costs, accumulate=d > 0)
            if want:
                g = self._buf('aux_g%d' % d, rows, F)
                ops.affine_rows(self.inner_hid(d), 2 * F, coef[:B], coef[B:2 * B], coef[2 * B:], mask, B, g, F, rows, F)
                self.inner_grad[d] = g
        return costs[0]

    _SAVE = (('gi', 4), ('go', 4), ('ci', 1), ('co', 1), ('r', 2), ('xhi', 1), ('xho', 1), ('sc', 12), ('a', 4))
    _BWD = (('dr', 2), ('dta', 8), ('doi', 4), ('dsc', 12), ('douti', 1), ('douto', 1))

    def _desc(self, T, B, mask, backward):
        N, F = self.N, self.F
        desc = _lib.PrjHyperDesc()
        desc.T, desc.N, desc.F, desc.ndir, desc.Btot = T, N, F, self.ndir, B
        desc.grad_clip = self.clip
        desc.mask = mask.data_ptr(); desc.mask_stride_b = mask.stride(0); desc.mask_stride_t = mask.stride(1)
        for d, dp in enumerate(self.dirs):
            c = self.ctx['d'][d]
            Dd = desc.dir[d]
            Dd.reverse = 1 if self.reverse[d] else 0
            for f, key in (('s', 's'), ('ai', 'ai'), ('bi_c', 'bic'), ('ao', 'ao'), ('bo_c', 'boc')):
                setattr(Dd, f, dp.p(key).data_ptr())
            for f in ('inner_in', 'outer_in', 'Wf', 'W3', 'rsi', 'rso') + tuple(n for n, _ in self._SAVE):
                setattr(Dd, f, c[f].data_ptr())
            if backward:
                for f in tuple(n for n, _ in self._BWD):
                    setattr(Dd, f, c[f].data_ptr())
        return desc

    def _run(self, desc, B, backward):
        L = _lib.lib()
        need = L.asr_prjhyper_ws_bytes(min(B, 64), self.N, self.F, self.ndir)
        ws = self.net._workspace(('prjhyper', self.idx), need)
        fn = L.asr_prjhyper_bwd if backward else L.asr_prjhyper_fwd
        for b0 in range(0, B, 64):                      # batch rows are independent: <= 64 rows per call
            desc.b0 = b0; desc.B = min(64, B - b0)
            check(fn(C.byref(desc), ws.data_ptr(), ws.numel(), torch.cuda.current_stream().cuda_stream),
                  'asr_prjhyper_%s' % ('bwd' if backward else 'fwd'))
        # the persistent launch's sync block heads the workspace: Network.check_status reads its abort word
        self.net._last_groups = getattr(self.net, '_last_groups', {})
        self.net._last_groups[('prjhyper', self.idx)] = 1

    def forward(self, X, T, B, mask, cond, train):
        N, F = self.N, self.F
        rows = T * B
        Y = self._buf('Y', rows, self.out_dim)
        self.ctx = ctx = dict(X=X, T=T, B=B, mask=mask, Y=Y, d=[])
        for d, dp in enumerate(self.dirs):
            c = {}
            c['inner_in'] = ops.gemm(X, dp.p('Wix'), self._buf('iin%d' % d, rows, 4 * N), bias=dp.p('bi'))
            c['outer_in'] = ops.gemm(X, dp.p('Wox'), self._buf('oin%d' % d, rows, 4 * N))
            # both projections folded into their consumers (step_prjhyper.hip):
            #   Wf = [[Wip.Wio[:F], 0], [Wop.Wio[F:], Wop.Woo]] (2N,8N),  W3 = Wip.S (N,12N)
            Wf = self._buf('Wf%d' % d, 2 * N, 8 * N)
            Wf[:N, 4 * N:].zero_()
            ops.gemm(dp.p('Wip'), dp.p('Wio')[:F], Wf[:N, :4 * N])
            ops.gemm(dp.p('Wop'), dp.p('Wio')[F:], Wf[N:, :4 * N])
            ops.gemm(dp.p('Wop'), dp.p('Woo'), Wf[N:, 4 * N:])
            c['Wf'] = Wf
            c['W3'] = ops.gemm(dp.p('Wip'), dp.p('S'), self._buf('W3%d' % d, N, 12 * N))
            c['hid'] = self._buf('hid%d' % d, rows, 2 * F)
            c['rsi'] = self._buf('rsi%d' % d, rows, 1); c['rso'] = self._buf('rso%d' % d, rows, 1)
            for nm, w in self._SAVE:
                c[nm] = self._buf('%s%d' % (nm, d), rows, w * N)
            ctx['d'].append(c)
        self._run(self._desc(T, B, mask, False), B, False)
        for d, dp in enumerate(self.dirs):
            c = ctx['d'][d]
            ops.gemm(c['r'][:, :N], dp.p('Wip'), c['hid'][:, :F])                  # :364 inner_hid, all T at once
            ops.gemm(c['r'][:, N:], dp.p('Wop'), c['hid'][:, F:])                  # :410 outer_hid
            Y[:, d * F:(d + 1) * F].copy_(c['hid'][:, F:])                         # SliceLayer(num_factors:) + ConcatLayer
        return Y

    def backward(self, dY, need_dx=True):
        net = self.net; ctx = self.ctx; N, F = self.N, self.F
        X, T, B, mask = ctx['X'], ctx['T'], ctx['B'], ctx['mask']
        rows = T * B
        for d in range(self.ndir):
            c = ctx['d'][d]
            c['dhid'] = self._buf('dhid%d' % d, rows, 2 * F)
            if self.inner_grad[d] is None:
                c['dhid'][:, :F].zero_()
            else:
                c['dhid'][:, :F].copy_(self.inner_grad[d]); self.inner_grad[d] = None
            c['dhid'][:, F:].copy_(dY[:, d * F:(d + 1) * F])
            for nm, w in self._BWD:
                c[nm] = self._buf('%s%d' % (nm, d), rows, w * N)
            dp = self.dirs[d]                           # the external part of d r: d hid . W_prj^T, all T at once
            ops.gemm(c['dhid'][:, :F], dp.p('Wip'), c['dr'][:, :N], tb=True)
            ops.gemm(c['dhid'][:, F:], dp.p('Wop'), c['dr'][:, N:], tb=True)
        self._run(self._desc(T, B, mask, True), B, True)
        dX = self._buf('dX', rows, self.D) if need_dx else None
        if need_dx:
            for d, dp in enumerate(self.dirs):
                c = ctx['d'][d]
                ops.gemm(c['dta'][:, :4 * N], dp.p('Wix'), dX, tb=True, beta=0.0 if d == 0 else 1.0)
                ops.gemm(c['doi'], dp.p('Wox'), dX, tb=True, beta=1.0)
        with net.deferred():
            self._param_grads(ctx, rows, T, B)
        return dX

    def _param_grads(self, ctx, rows, T, B):
        N, F = self.N, self.F
        X = ctx['X']
        n1 = (T - 1) * B
        for d, dp in enumerate(self.dirs):
            c = ctx['d'][d]
            r, dta, dhid = c['r'], c['dta'], c['dhid']
            dii = dta[:, :4 * N]
            if not self.reverse[d]:
                sl_prev, sl_cur = slice(0, n1), slice(B, rows)
            else:
                sl_prev, sl_cur = slice(B, rows), slice(0, n1)
            ops.gemm(X, dii, dp.g('Wix'), ta=True)
            ops.colreduce(dii, dp.g('bi'), rows, 4 * N, ldx=8 * N)
            ops.gemm(X, c['doi'], dp.g('Wox'), ta=True)
            # gradients of the folded matrices, then the product rule back to the reference's factors
            dWf = self._buf('dWf%d' % d, 2 * N, 8 * N)
            if n1:
                ops.gemm(r[sl_prev], dta[sl_cur], dWf, ta=True)
            else:                                       # T == 1: no recurrent step ran
                dWf.zero_()
            dW3 = ops.gemm(r[:, :N], c['dsc'], self._buf('dW3%d' % d, N, 12 * N), ta=True)
            Wip, Wop, Wio, Woo, S = dp.p('Wip'), dp.p('Wop'), dp.p('Wio'), dp.p('Woo'), dp.p('S')
            gWip, gWop, gWio = dp.g('Wip'), dp.g('Wop'), dp.g('Wio')
            ops.gemm(r[:, :N], dhid[:, :F], gWip, ta=True)                         # hid as an output: inner, outer
            ops.gemm(r[:, N:], dhid[:, F:], gWop, ta=True)
            ops.gemm(dWf[:N, :4 * N], Wio[:F], gWip, tb=True, beta=1.0)            # Wf[:N,:4N] = Wip.Wio[:F]
            ops.gemm(Wip, dWf[:N, :4 * N], gWio[:F], ta=True)
            ops.gemm(dWf[N:, :4 * N], Wio[F:], gWop, tb=True, beta=1.0)            # Wf[N:,:4N] = Wop.Wio[F:]
            ops.gemm(Wop, dWf[N:, :4 * N], gWio[F:], ta=True)
            ops.gemm(dWf[N:, 4 * N:], Woo, gWop, tb=True, beta=1.0)                # Wf[N:,4N:] = Wop.Woo
            ops.gemm(Wop, dWf[N:, 4 * N:], dp.g('Woo'), ta=True)
            ops.gemm(dW3, S, gWip, tb=True, beta=1.0)                              # W3 = Wip.S
            ops.gemm(Wip, dW3, dp.g('S'), ta=True)
            ops.colreduce(c['dsc'], dp.g('s'), rows, 12 * N)
            ops.colreduce(c['douti'], dp.g('ai'), rows, N, y=c['xhi'])
            ops.colreduce(c['douti'], dp.g('bic'), rows, N)
            ops.colreduce(c['douto'], dp.g('ao'), rows, N, y=c['xho'])
            ops.colreduce(c['douto'], dp.g('boc'), rows, N)

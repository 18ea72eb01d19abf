"""The LSTM layer classes of libs/lasagne_libs/recurrent_layers.py (20 classes,
one variant table, asr_amd/specs.py REC_CLASSES) on the persistent LSTM-family scan kernel.

Per direction the host side prepares, with one kernel each:
  conditioning prologue  s = pool_t act(src.W_cond + b)  (src = the layer's own input or the external
                         i-vector input), alpha_in / alpha_hid / beta = s.W + b   (:944-952, :1209-1217)
  input side             xin = [LN]( x.W_in [+ beta] ) [* alpha_in] [+ beta | + b]  for all T at once
  recurrent matrix       W' = W_hid_prj . W_hid folded (N x 4N) when the class projects
and the scan kernel does the rest (asr_amd/csrc/scan_lstm.hip).  Parameter gradients run deferred on
the side stream; for own-input conditioning the pooled-feature gradient is on the critical path
because it flows into dX.
"""
import ctypes as C
import torch

from . import _lib, ops, specs
from .engine import DirParams

G4R = specs.GATES_REC


def _mk_rec(store, prefix, cls, D, N, P, F, I, peepholes=False, learn_init=False):
    v = specs.REC_CLASSES[cls]
    dp = DirParams(store, prefix)
    K = P if P else N
    dp.add('Wx', (D, 4 * N)); dp.add('Wh', (K, 4 * N))
    if v['bias']:
        dp.add('b', (4 * N,))
    if peepholes:                  # (N,4) per unit [w_ci, w_cf, 0, w_co]: the kernel's one 16-byte read per lane
        dp.add('pe', (4 * N,))
    if learn_init:
        dp.add('c_init', (1, N)); dp.add('h_init', (1, K))
    if P:
        dp.add('Wp', (N, P))
    if v['ln'] == 'static':
        dp.add('a_in', (4 * N,)); dp.add('a_hid', (4 * N,))
        if v['ln_beta']:
            dp.add('b_in', (4 * N,)); dp.add('b_hid', (4 * N,))
    if v['cell_ln']:
        dp.add('a_c', (N,)); dp.add('b_c', (N,))
    if v['cond']:
        cin = D if v['cond'] == 'own' else I
        dp.add('Wc', (cin, F)); dp.add('bc', (F,))
        for nm in ('ai', 'ah', 'bt'):
            dp.add('W' + nm, (F, 4 * N))
        dp.add('bai', (4 * N,)); dp.add('bah', (4 * N,))
        if not v['bias']:
            dp.add('bbt', (4 * N,))
    # reference order / names
    table = {'W_hid_prj': ('Wp', 'same'), 'W_cond_prj': ('Wc', 'same'), 'b_cond_prj': ('bc', 'same'),
             'W_alpha_cell': ('a_c', 'same'), 'W_beta_cell': ('b_c', 'same'), 'num_factors': ('b_c', 'same'),
             'W_beta_in': ('b_in', 'perm4'), 'W_beta_hid': ('b_hid', 'perm4'),
             'b_alpha_in': ('bai', 'perm4'), 'b_alpha_hid': ('bah', 'perm4'),
             'W_beta': ('Wbt', 'perm4'), 'b_beta': ('bbt', 'perm4'),
             'cell_init': ('c_init', 'same'), 'hid_init': ('h_init', 'same')}
    for name, shape, _ in specs.rec_params(cls, D, N, F if v['cond'] else (P or None), P or None, I,
                                           peepholes=peepholes, learn_init=learn_init):
        done = False
        for g, gn in enumerate(G4R):
            if name == 'W_in_to_' + gn:
                dp.map(name, 'Wx', 'gatecols', g); done = True
            elif name == 'W_hid_to_' + gn:
                dp.map(name, 'Wh', 'gatecols', g); done = True
            elif name == 'b_' + gn:
                dp.map(name, 'b', 'gatevec', g); done = True
        for g, gn in ((0, 'ingate'), (1, 'forgetgate'), (3, 'outgate')):
            if name == 'W_cell_to_' + gn:
                dp.map(name, 'pe', 'gatevec', g); done = True
        if done:
            continue
        if name == 'W_alpha_in':
            dp.map(name, 'Wai' if v['cond'] else 'a_in', 'perm4')
        elif name == 'W_alpha_hid':
            dp.map(name, 'Wah' if v['cond'] else 'a_hid', 'perm4')
        else:
            key, kind = table[name]
            dp.map(name, key, kind)
    return dp


class RecPair(object):
    """A forward and optionally a backward direction of one recurrent_layers.py class."""
    stateful = False

    def __init__(self, net, idx, cls, D, N, clip, ndir, num_factors=None, num_prjs=None, cond_dim=0,
                 peepholes=False, learn_init=False, only_return_final=False, only_return_hidden=True):
        """peepholes / learn_init / only_return_final / only_return_hidden: the classes' options
        (libs/lasagne_libs/recurrent_layers.py:52-57), defaults as there.  A layer with only_return_final returns
        (B, out_dim) rows and can only be the caller's last step (Network chains (T*B, .) sequences)."""
        v = specs.REC_CLASSES[cls]
        specs.rec_check_options(cls, peepholes, learn_init)
        self.peepholes = bool(peepholes); self.learn_init = bool(learn_init)
        self.only_return_final = bool(only_return_final); self.only_return_hidden = bool(only_return_hidden)
        self.opt = self.peepholes or self.learn_init or not self.only_return_hidden
        self.net = net; self.idx = idx; self.kind = 'rec'; self.cls = cls; self.v = v
        self.D = D; self.N = N; self.clip = float(clip); self.ndir = ndir
        self.num_factors = num_factors; self.num_prjs = num_prjs
        self.P = specs.rec_out_width(cls, N, num_factors, num_prjs)
        self.F = num_factors if v['cond'] else 0
        self.I = cond_dim
        self.wdir = (self.P or N) + (0 if self.only_return_hidden else N)     # columns of one direction's output
        self.out_dim = ndir * self.wdir
        self.ln_kind = {'none': 0, 'static': 1, 'sample': 2}[v['ln']]     # how alpha is indexed
        self.grp = 4 if v['per_gate'] else 1
        self.cell_ln = 1 if v['cell_ln'] else 0
        self.dirs = [_mk_rec(net.store, 'L%d.%d.' % (idx, d), cls, D, N, self.P, self.F, cond_dim, self.peepholes,
                             self.learn_init) for d in range(ndir)]
        self.reverse = [False, True][:ndir]

    def spec_per_dir(self):
        v = self.v
        return specs.rec_params(self.cls, self.D, self.N, self.F if v['cond'] else (self.P or None),
                                self.P or None, self.I, peepholes=self.peepholes, learn_init=self.learn_init)

    def describe(self):
        out = {}
        for d, nm in enumerate(('fwd', 'bwd')):
            if d >= self.ndir:
                out[nm] = None
                continue
            args = dict(variant=dict(self.v), num_units=self.N, out_width=self.P, grad_clipping=self.clip,
                        backwards=self.reverse[d], cls=self.cls)
            for k, dflt in (('peepholes', False), ('learn_init', False), ('only_return_final', False),
                            ('only_return_hidden', True)):       # options only when set: the frozen fixtures of
                if getattr(self, k) != dflt:                     # tests/golden/ describe default layers without them
                    args[k] = getattr(self, k)
            out[nm] = dict(kind='rec', args=args, params=[r[0] for r in self.dirs[d].refmap])
        return out

    def get_sample_feat(self, d=0):
        """recurrent_layers.py:1337-1338 (pooled conditioning feature of the last forward)."""
        return self.ctx['d'][d]['s']

    def _buf(self, name, rows, cols):
        return self.net.arena.mat('L%d.%s' % (self.idx, name), rows, cols)

    # ---------------------------------------------------------------------------------------------
    def _cond_fwd(self, d, dp, c, X, T, B, mask, cond):
        v = self.v; F = self.F; N = self.N
        if v['cond'] == 'own':
            src = X; rows = T * B; sb, st, Tc = F, B * F, T
        else:
            Tc = cond.shape[1]
            src = cond.view(B * Tc, self.I); rows = B * Tc; sb, st = Tc * F, F
        mode = 0 if v['pool'] == 'mean_tanh' else 1
        pre = self._buf('pre%d' % d, rows, F)
        ops.gemm(src, dp.p('Wc'), pre)
        ops.bias_act(pre, dp.p('bc'), rows, F, 2 if mode == 0 else 1)
        s = self._buf('s%d' % d, B, F)
        ops.seq_pool_fwd(pre, sb, st, mask, B, Tc, F, mode, s)
        al_in = self._buf('al_in%d' % d, B, 4 * N); al_hid = self._buf('al_hid%d' % d, B, 4 * N)
        bt = self._buf('bt%d' % d, B, 4 * N)
        ops.gemm(s, dp.p('Wai'), al_in, bias=dp.p('bai'))
        ops.gemm(s, dp.p('Wah'), al_hid, bias=dp.p('bah'))
        ops.gemm(s, dp.p('Wbt'), bt, bias=dp.p('b') if v['bias'] else dp.p('bbt'))
        c.update(src=src, pre=pre, s=s, al_in=al_in, al_hid=al_hid, bt=bt, pool=(sb, st, Tc, mode))

    def prepare(self, train, T, B, mask, cond):
        """Weight-only forward work (engine.Network.prepare_ahead): folded / packed recurrent matrices and, for the
        classes conditioned on an external input (i-vector), the whole conditioning prologue."""
        net = self.net; N = self.N; P = self.P; v = self.v
        L = _lib.lib()
        out = []
        for d, dp in enumerate(self.dirs):
            c = {}
            if v['cond'] and v['cond'] != 'own':
                self._cond_fwd(d, dp, c, None, T, B, mask, cond)
            if P:
                Wf = self._buf('Wf%d' % d, N, 4 * N)
                ops.gemm(dp.p('Wp'), dp.p('Wh'), Wf)
            else:
                Wf = dp.p('Wh')
            c['wpk'] = ops.lstm_pack(Wf, N, 0, net.arena.get('L%d.wpk%d' % (self.idx, d), L.asr_lstm_pack_floats(N, 0)))
            if self.learn_init:        # T.dot(ones, cell_init) / T.dot(ones, hid_init), recurrent_layers.py:246-248
                ones = net.arena.mat('ones_col', B, 1); ones.fill_(1.0)
                c['c0'] = ops.gemm(ones, dp.p('c_init'), self._buf('c0_%d' % d, B, N))
                c['h0'] = ops.gemm(ones, dp.p('h_init'), self._buf('h0_%d' % d, B, P or N))
                if P:                  # the projected initial row enters the recurrence through W_h, not through W'
                    c['a0'] = ops.gemm(c['h0'], dp.p('Wh'), self._buf('a0_%d' % d, B, 4 * N))
            if train:
                pm = 1 if self.opt else L.asr_lstm_bwd_pack_mode(N, self.ndir, self.ln_kind + (2 if self.grp == 4 else 0))
                c['wpkb'] = ops.lstm_pack(Wf, N, pm, net.arena.get('L%d.wpkb%d' % (self.idx, d),
                                                                   L.asr_lstm_pack_floats(N, pm)))
            out.append(c)
        return out

    def forward(self, X, T, B, mask, cond, train):
        net = self.net; N = self.N; P = self.P; v = self.v; rows = T * B
        Pout = P or N
        W = self.wdir
        Y = self._buf('Y', rows, self.out_dim)
        desc = _lib.LstmScanDesc()
        desc.T, desc.B, desc.N, desc.ndir = T, B, N, self.ndir
        desc.ln_kind, desc.cell_ln, desc.grad_clip = self.ln_kind + (2 if self.grp == 4 else 0), self.cell_ln, self.clip
        desc.mask = mask.data_ptr(); desc.mask_stride_b = mask.stride(0); desc.mask_stride_t = mask.stride(1)
        self.ctx = ctx = dict(X=X, T=T, B=B, mask=mask, cond=cond, Y=Y, d=[])
        L = _lib.lib()
        prep = net.take_prepared(self) or self.prepare(train, T, B, mask, cond)
        for d, dp in enumerate(self.dirs):
            c = prep[d]
            if v['cond'] == 'own':
                self._cond_fwd(d, dp, c, X, T, B, mask, cond)
            xin = self._buf('xin%d' % d, rows, 4 * N)
            if self.ln_kind == 0:
                ops.gemm(X, dp.p('Wx'), xin, bias=dp.p('b'))
            else:
                ops.gemm(X, dp.p('Wx'), xin)
                xh = self._buf('xhat_in%d' % d, rows, 4 * N); rs = self._buf('rs_in%d' % d, rows, self.grp)
                if self.ln_kind == 1:
                    ops.ln_rows_fwd(xin, 4 * N, dp.p('a_in'), xin, beta=dp.p('b_in') if v['ln_beta'] else None,
                                    bias=dp.p('b'), xhat=xh, rstd=rs, groups=self.grp)
                elif v['beta'] == 'pre':
                    ops.ln_rows_fwd(xin, 4 * N, c['al_in'], xin, pre=c['bt'], pmod=B, amod=B, xhat=xh, rstd=rs,
                                    groups=self.grp)
                else:
                    ops.ln_rows_fwd(xin, 4 * N, c['al_in'], xin, beta=c['bt'], amod=B, xhat=xh, rstd=rs,
                                    groups=self.grp)
                c.update(xhat_in=xh, rs_in=rs)
            Dd = desc.dir[d]
            Dd.reverse = 1 if self.reverse[d] else 0
            Dd.xin = xin.data_ptr(); Dd.wpack = c['wpk'].data_ptr()
            if self.ln_kind == 1:
                Dd.alpha_hid = dp.p('a_hid').data_ptr()
                Dd.beta_hid = dp.p('b_hid').data_ptr() if v['ln_beta'] else None
            elif self.ln_kind == 2:
                Dd.alpha_hid = c['al_hid'].data_ptr()
            if self.cell_ln:
                Dd.alpha_c = dp.p('a_c').data_ptr(); Dd.beta_c = dp.p('b_c').data_ptr()
            R = self._buf('R%d' % d, rows, N) if P else Y[:, d * W:d * W + N]
            c['R'] = R
            Dd.r_out = R.data_ptr(); Dd.ld_r = R.stride(0)
            c['cst'] = self._buf('cst%d' % d, rows, N); Dd.cst = c['cst'].data_ptr()
            if self.peepholes:
                Dd.peep = dp.p('pe').data_ptr()
            if self.learn_init:
                Dd.c0 = c['c0'].data_ptr()
                if P:
                    Dd.a0 = c['a0'].data_ptr(); Dd.a0_until_valid = 1
                else:
                    Dd.r0 = c['h0'].data_ptr()
            if train:
                c['gates'] = self._buf('gates%d' % d, rows, 4 * N); Dd.gates = c['gates'].data_ptr()
            if self.ln_kind:
                c['ahat'] = self._buf('ahat%d' % d, rows, 4 * N); c['rs_a'] = self._buf('rs_a%d' % d, rows, self.grp)
                Dd.ahat = c['ahat'].data_ptr(); Dd.rs_a = c['rs_a'].data_ptr()
            if self.cell_ln:
                c['chat'] = self._buf('chat%d' % d, rows, N); c['rs_c'] = self._buf('rs_c%d' % d, rows, 1)
                Dd.chat = c['chat'].data_ptr(); Dd.rs_c = c['rs_c'].data_ptr()
            ctx['d'].append(c)
        net.scan_lstm(self, desc, backward=False)
        for d, dp in enumerate(self.dirs):
            c = ctx['d'][d]
            Yd = Y[:, d * W:d * W + Pout]
            if P:
                ops.gemm(c['R'], dp.p('Wp'), Yd)
                if self.learn_init and self.reverse[d]:
                    # a row's leading masked steps of the reverse scan (its padding) hold hid_init itself, which is no
                    # r.W_p: r is zero there, so Y += (1 - mask) * hid_init   (T.switch(mask_n, hid, hid_previous), :239)
                    one = self._buf('one_bp', B, Pout); one.fill_(1.0)
                    nh0 = ops.gemm(net.arena.mat('ones_col', B, 1), dp.p('h_init'), self._buf('nh0_%d' % d, B, Pout),
                                   alpha=-1.0)
                    ops.affine_rows(Yd, Y.stride(0), one, c['h0'], nh0, mask, B, Yd, Y.stride(0), rows, Pout)
            if not self.only_return_hidden:          # cell_out beside hid_out (:275-288)
                Y[:, d * W + Pout:(d + 1) * W].copy_(c['cst'])
        if self.only_return_final:                   # hid_out[-1] in scan order (:265-266): time T-1, or 0 going backwards
            Yf = self._buf('Yfin', B, self.out_dim)
            for d in range(self.ndir):
                r0 = 0 if self.reverse[d] else (T - 1) * B
                Yf[:, d * W:(d + 1) * W].copy_(Y[r0:r0 + B, d * W:(d + 1) * W])
            return Yf
        return Y

    # ---------------------------------------------------------------------------------------------
    def backward(self, dY, need_dx=True):
        net = self.net; ctx = self.ctx; N = self.N; P = self.P; v = self.v
        X, T, B, mask = ctx['X'], ctx['T'], ctx['B'], ctx['mask']
        rows = T * B
        Pout = P or N
        W = self.wdir
        later = []
        if self.only_return_final:                   # the gradient of hid_out[-1] (cell_out[-1]) sits on one time step
            dYf = self._buf('dYfull', rows, self.out_dim); dYf.zero_()
            for d in range(self.ndir):
                r0 = 0 if self.reverse[d] else (T - 1) * B
                dYf[r0:r0 + B, d * W:(d + 1) * W].copy_(dY[:, d * W:(d + 1) * W])
            dY = dYf
        desc = _lib.LstmScanDesc()
        desc.T, desc.B, desc.N, desc.ndir = T, B, N, self.ndir
        desc.ln_kind, desc.cell_ln, desc.grad_clip = self.ln_kind + (2 if self.grp == 4 else 0), self.cell_ln, self.clip
        desc.mask = mask.data_ptr(); desc.mask_stride_b = mask.stride(0); desc.mask_stride_t = mask.stride(1)
        for d, dp in enumerate(self.dirs):
            c = ctx['d'][d]
            dYd = dY[:, d * W:d * W + Pout]
            c['dYd'] = dYd
            if P:
                dR = self._buf('dR%d' % d, rows, N)
                ops.gemm(dYd, dp.p('Wp'), dR, tb=True)
                later.append(lambda dp=dp, c=c, dYd=dYd: ops.gemm(c['R'], dYd, dp.g('Wp'), ta=True))
            else:
                dR = dYd
            Dd = desc.dir[d]
            Dd.reverse = 1 if self.reverse[d] else 0
            Dd.wpack = c['wpkb'].data_ptr()
            if self.ln_kind == 1:
                Dd.alpha_hid = dp.p('a_hid').data_ptr()
            elif self.ln_kind == 2:
                Dd.alpha_hid = c['al_hid'].data_ptr()
            if self.cell_ln:
                Dd.alpha_c = dp.p('a_c').data_ptr(); Dd.beta_c = dp.p('b_c').data_ptr()
            Dd.gates = c['gates'].data_ptr(); Dd.cst = c['cst'].data_ptr()
            if self.ln_kind:
                Dd.ahat = c['ahat'].data_ptr(); Dd.rs_a = c['rs_a'].data_ptr()
                c['dA'] = self._buf('dA%d' % d, rows, 4 * N); Dd.dA = c['dA'].data_ptr()
            if self.cell_ln:
                Dd.chat = c['chat'].data_ptr(); Dd.rs_c = c['rs_c'].data_ptr()
                c['dOut'] = self._buf('dOut%d' % d, rows, N); Dd.dOut = c['dOut'].data_ptr()
            Dd.dr_ext = dR.data_ptr(); Dd.ld_dr = dR.stride(0)
            c['dG'] = self._buf('dG%d' % d, rows, 4 * N); Dd.dG = c['dG'].data_ptr()
            if not self.only_return_hidden:
                dC = dY[:, d * W + Pout:(d + 1) * W]
                Dd.dc_ext = dC.data_ptr(); Dd.ld_dc = dC.stride(0)
            if self.peepholes:
                c['dpe'] = self._buf('dpe%d' % d, B, 4 * N)
                Dd.peep = dp.p('pe').data_ptr(); Dd.dpeep = c['dpe'].data_ptr()
            if self.learn_init:
                Dd.c0 = c['c0'].data_ptr()
                c['dc0'] = self._buf('dc0_%d' % d, B, N); Dd.dc0 = c['dc0'].data_ptr()
                if P:
                    c['da0'] = self._buf('da0_%d' % d, B, 4 * N); Dd.da0 = c['da0'].data_ptr()
                else:
                    c['dr0'] = self._buf('dr0_%d' % d, B, N); Dd.dr0 = c['dr0'].data_ptr()
        net.scan_lstm(self, desc, backward=True)
        own = v['cond'] == 'own'
        dX = self._buf('dX', rows, self.D) if (need_dx or False) else None
        for d, dp in enumerate(self.dirs):
            c = ctx['d'][d]
            dG = c['dG']
            if self.ln_kind == 0:
                dXp = dG
            else:
                dXp = self._buf('dXp%d' % d, rows, 4 * N)
                al = dp.p('a_in') if self.ln_kind == 1 else c['al_in']
                ops.ln_rows_bwd(dG, 4 * N, c['xhat_in'], c['rs_in'], al, dXp, amod=0 if self.ln_kind == 1 else B,
                                groups=self.grp)
            c['dXp'] = dXp
            if need_dx:
                ops.gemm(dXp, dp.p('Wx'), dX, tb=True, beta=0.0 if d == 0 else 1.0)
            if v['cond']:
                if own and need_dx:        # the pooled-feature gradient flows into dX: critical path
                    self._cond_bwd(d, dp, c, rows, B)
                    ops.gemm(c['dpre'], dp.p('Wc'), dX, tb=True, beta=1.0)
                    later.append(lambda d=d, dp=dp, c=c: self._cond_param_grads(d, dp, c, B))
                else:
                    later.append(lambda d=d, dp=dp, c=c: (self._cond_bwd(d, dp, c, rows, B),
                                                          self._cond_param_grads(d, dp, c, B)))
            later.append(lambda d=d, dp=dp, c=c: self._param_grads(d, dp, c, X, rows, T, B))
        with net.deferred():
            for fn in later:
                fn()
        return dX

    def _cond_bwd(self, d, dp, c, rows, B):
        """d alpha_in, d alpha_hid, d beta (B,4N) -> d s (B,F) -> d pre (layout of pre)."""
        N = self.N; F = self.F; v = self.v
        dG = c['dG']
        dal_in = self._buf('dal_in%d' % d, B, 4 * N); dal_hid = self._buf('dal_hid%d' % d, B, 4 * N)
        dbt = self._buf('dbt%d' % d, B, 4 * N)
        ops.colreduce(dG, dal_in, rows, 4 * N, y=c['xhat_in'], gmod=B)
        ops.colreduce(dG, dal_hid, rows, 4 * N, y=c['ahat'], gmod=B)
        ops.colreduce(c['dXp'] if v['beta'] == 'pre' else dG, dbt, rows, 4 * N, gmod=B)
        ds = self._buf('ds%d' % d, B, F)
        for k, (nm, dv) in enumerate((('ai', dal_in), ('ah', dal_hid), ('bt', dbt))):
            ops.gemm(dv, dp.p('W' + nm), ds, tb=True, beta=0.0 if k == 0 else 1.0)
        fr = getattr(self, 'feat_reg', 0.0)
        if fr:
            # + fr * d(-mean_j var_b s_bj)/ds = fr * (-2/(B F)) (I - 11^T/B) s: one more GEMM with the
            # constant centring matrix (scripts/dln/ted/exp_prj_lstm_dln.py:120-131)
            cm = self._centre.get(B) if hasattr(self, '_centre') else None
            if cm is None:
                import numpy as np
                if not hasattr(self, '_centre'):
                    self._centre = {}
                cm = torch.from_numpy((np.eye(B) - 1.0 / B).astype(np.float32)).to(self.net.device)
                self._centre[B] = cm
            ops.gemm(cm, c['s'], ds, alpha=fr * (-2.0 / (B * F)), beta=1.0)
        sb, st, Tc, mode = c['pool']
        dpre = self._buf('dpre%d' % d, c['pre'].shape[0], F)
        ops.seq_pool_bwd(ds, c['s'], c['pre'], sb, st, self.ctx['mask'], B, Tc, F, mode, dpre)
        c.update(dal_in=dal_in, dal_hid=dal_hid, dbt=dbt, dpre=dpre)

    def _cond_param_grads(self, d, dp, c, B):
        N = self.N; F = self.F; v = self.v
        s = c['s']
        for nm, dv in (('ai', c['dal_in']), ('ah', c['dal_hid']), ('bt', c['dbt'])):
            ops.gemm(s, dv, dp.g('W' + nm), ta=True)
        ops.colreduce(c['dal_in'], dp.g('bai'), B, 4 * N)
        ops.colreduce(c['dal_hid'], dp.g('bah'), B, 4 * N)
        ops.colreduce(c['dbt'], dp.g('b') if v['bias'] else dp.g('bbt'), B, 4 * N)
        ops.gemm(c['src'], c['dpre'], dp.g('Wc'), ta=True)
        ops.colreduce(c['dpre'], dp.g('bc'), c['dpre'].shape[0], F)

    def _param_grads(self, d, dp, c, X, rows, T, B):
        N = self.N; P = self.P; v = self.v
        dG = c['dG']; dA = c['dA'] if self.ln_kind else dG
        R = c['R']
        dWf = self._buf('dWf%d' % d, N, 4 * N) if P else dp.g('Wh')
        n1 = (T - 1) * B
        if not self.reverse[d]:
            Ra, dAb = R[:n1], dA[B:]
        else:
            Ra, dAb = R[B:], dA[:n1]
        ops.gemm(Ra, dAb, dWf, ta=True)
        if P:
            ops.gemm(dp.p('Wp'), dWf, dp.g('Wh'), ta=True)
            ops.gemm(dWf, dp.p('Wh'), dp.g('Wp'), tb=True, beta=1.0)
        if self.peepholes:
            ops.colreduce(c['dpe'], dp.g('pe'), B, 4 * N)
        if self.learn_init:
            ops.colreduce(c['dc0'], dp.g('c_init').view(-1), B, N)
            first = dA[(T - 1) * B:] if self.reverse[d] else dA[:B]       # the first scan step's rows
            srow = self._buf('srow%d' % d, 1, 4 * N)
            if P:
                # hid_init enters through a0 = h0.W_h at each row's first unmasked step (da0), and IS the output at the
                # masked steps before it (reverse direction: the row's padding)
                ops.gemm(c['h0'], c['da0'], dp.g('Wh'), ta=True, beta=1.0)
                ops.colreduce(c['da0'], srow.view(-1), B, 4 * N)
                ops.gemm(srow, dp.p('Wh'), dp.g('h_init'), tb=True)
                if self.reverse[d]:
                    Pw = P
                    zero = self._buf('zero_bp', B, Pw); zero.zero_()
                    one = self._buf('one_bp', B, Pw); one.fill_(1.0)
                    mone = self._buf('mone_bp', B, Pw); mone.fill_(-1.0)
                    E = self._buf('invmask%d' % d, rows, Pw)
                    dYd = c['dYd']
                    ops.affine_rows(dYd, dYd.stride(0), zero, one, mone, self.ctx['mask'], B, E, Pw, rows, Pw)
                    ops.colreduce(dYd, dp.g('h_init').view(-1), rows, Pw, y=E, accumulate=True, ldx=dYd.stride(0), ldy=Pw)
            else:
                # gradient wrt r0 = what passed through masked steps + the first scan step's da.W_h^T
                ops.gemm(c['h0'], first, dp.g('Wh'), ta=True, beta=1.0)
                ops.colreduce(first, srow.view(-1), B, 4 * N)
                ops.gemm(srow, dp.p('Wh'), dp.g('h_init'), tb=True)
                ops.colreduce(c['dr0'], dp.g('h_init').view(-1), B, N, accumulate=True)
        if v['bias'] and not v['cond']:
            ops.colreduce(dG, dp.g('b'), rows, 4 * N)
        if self.cell_ln:
            ops.colreduce(c['dOut'], dp.g('a_c'), rows, N, y=c['chat'])
            ops.colreduce(c['dOut'], dp.g('b_c'), rows, N)
        if self.ln_kind == 1:
            ops.colreduce(dG, dp.g('a_in'), rows, 4 * N, y=c['xhat_in'])
            ops.colreduce(dG, dp.g('a_hid'), rows, 4 * N, y=c['ahat'])
            if v['ln_beta']:
                ops.colreduce(dG, dp.g('b_in'), rows, 4 * N)
                ops.colreduce(dG, dp.g('b_hid'), rows, 4 * N)
        ops.gemm(X, c['dXp'], dp.g('Wx'), ta=True)

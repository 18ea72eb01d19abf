"""Second, independent restatement of the reference layers used ONLY to pin the numpy oracle:
torch float64, written step by step after the reference `get_output_for` bodies and differentiated
by torch autograd (the oracle's backward is hand-derived).  grad_clip is a custom autograd Function
with Theano's semantics (identity forward, clip of the incoming gradient backward)."""
import torch

EPS = float(torch.finfo(torch.float32).eps)


class _GradClip(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, c):
        ctx.c = c
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g.clamp(-ctx.c, ctx.c), None


def grad_clip(x, c):
    return _GradClip.apply(x, c) if c else x


def ln(x, alpha, beta=None):                       # libs/lhuclstm_layers.py:16-21
    out = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, keepdim=True, unbiased=False) + EPS)
    out = out * alpha
    if beta is not None:
        out = out + beta
    return out


def _cat(P, fmt, gates, dim=1):
    return torch.cat([P[fmt.format(g)] for g in gates], dim=dim)


G_L = ('ig', 'fg', 'c', 'og')
G_R = ('ingate', 'forgetgate', 'cell', 'outgate')


def _scan(step, seqs, init, backwards):
    T = seqs[0].shape[0]
    order = range(T - 1, -1, -1) if backwards else range(T)
    state = init
    outs = []
    for t in order:
        state = step(*[s[t] for s in seqs], *state)
        outs.append(state)
    return [torch.stack([o[k] for o in outs]) for k in range(len(init))]


def lstm_layer(P, x, mask, N, Pj=0, use_ln=False, clip=0, backwards=False, c0=None, h0=None,
               return_cell=False):
    """libs/lhuclstm_layers.py:73-130,172-218."""
    W_h = _cat(P, 'W_h_{}', G_L); W_x = _cat(P, 'W_x_{}', G_L); b = _cat(P, 'b_{}', G_L, 0)
    inp = x.transpose(0, 1)
    B = inp.shape[1]
    cell_init = torch.zeros(B, N, dtype=x.dtype) if c0 is None else c0
    hid_init = torch.zeros(B, Pj or N, dtype=x.dtype) if h0 is None else h0
    inp = inp @ W_x
    m = mask.transpose(0, 1).unsqueeze(-1)

    def step(input_n, mask_n, cell_previous, hid_previous):
        if use_ln:
            gates = ln(input_n, P['W_x_alpha'])
            gates = gates + ln(hid_previous @ W_h, P['W_h_alpha'])
        else:
            gates = input_n + hid_previous @ W_h
        gates = gates + b
        ig, fg, ci, og = [grad_clip(gates[:, k * N:(k + 1) * N], clip) for k in range(4)]
        ig = torch.sigmoid(ig); fg = torch.sigmoid(fg); ci = torch.tanh(ci); og = torch.sigmoid(og)
        cell = fg * cell_previous + ig * ci
        outcell = ln(cell, P['W_c_alpha'], P['W_c_beta']) if use_ln else cell
        outcell = grad_clip(outcell, clip)
        hid = og * torch.tanh(outcell)
        if Pj:
            hid = hid @ P['W_p']
        cell = torch.where(mask_n != 0, cell, cell_previous)
        hid = torch.where(mask_n != 0, hid, hid_previous)
        return [cell, hid]

    cell_out, hid_out = _scan(step, [inp, m], [cell_init, hid_init], backwards)
    hid_out = hid_out.transpose(0, 1); cell_out = cell_out.transpose(0, 1)
    if return_cell:
        return hid_out, cell_out            # iteration order (used for TBPTT state, forward only)
    if backwards:
        hid_out = hid_out.flip(1)
    return hid_out


def lhuc_layer(P, x, mask, ivec, N, npl, **kw):
    """libs/lhuclstm_layers.py:402-468,504-510."""
    hid = lstm_layer(P, x, mask, N, **kw)
    pred = ivec.transpose(0, 1)[0]
    for i in range(npl):
        pred = torch.relu(pred @ P['W_e_h_%d' % i] + P['b_e_h_%d' % i])
    sf = pred @ P['W_e_h_%d' % npl] + P['b_e_h_%d' % npl]
    return hid * torch.relu(sf).unsqueeze(1)


def prj_layer(P, x, mask, N, clip=0, backwards=False, kind='prj', cond=None):
    """recurrent_layers.py:404-546 (prj), :674-814 (ln_prj), :1202-1335 (icond_ln_prj)."""
    W_in = _cat(P, 'W_in_to_{}', G_R); W_hid = _cat(P, 'W_hid_to_{}', G_R)
    F = W_hid.shape[0]
    inp = x.transpose(0, 1)
    m = mask.transpose(0, 1).unsqueeze(-1)
    B = inp.shape[1]
    if kind == 'icond_ln_prj':
        li = torch.tanh(cond @ P['W_cond_prj'] + P['b_cond_prj'])
        li = (li * mask.unsqueeze(-1)).sum(1) / mask.sum(1, keepdim=True)
        alpha_in = li @ P['W_alpha_in'] + P['b_alpha_in']
        alpha_hid = li @ P['W_alpha_hid'] + P['b_alpha_hid']
        beta = li @ P['W_beta'] + P['b_beta']
        inp = inp @ W_in + beta
    elif kind == 'prj':
        inp = inp @ W_in + _cat(P, 'b_{}', G_R, 0)
    else:
        inp = inp @ W_in
        b_stacked = _cat(P, 'b_{}', G_R, 0)

    def step(input_n, mask_n, cell_previous, hid_previous):
        if kind == 'prj':
            gates = input_n + hid_previous @ W_hid
        elif kind == 'ln_prj':
            gates = ln(input_n, P['W_alpha_in'], P['W_beta_in'])
            gates = gates + ln(hid_previous @ W_hid, P['W_alpha_hid'], P['W_beta_hid'])
            gates = gates + b_stacked
        else:
            gates = ln(input_n, alpha_in) + ln(hid_previous @ W_hid, alpha_hid)
        ig, fg, ci, og = [grad_clip(gates[:, k * N:(k + 1) * N], clip) for k in range(4)]
        ig = torch.sigmoid(ig); fg = torch.sigmoid(fg); ci = torch.tanh(ci); og = torch.sigmoid(og)
        cell = fg * cell_previous + ig * ci
        outcell = ln(cell, P['W_alpha_cell'], P['W_beta_cell']) if kind == 'ln_prj' else cell
        outcell = grad_clip(outcell, clip)
        hid = (og * torch.tanh(outcell)) @ P['W_hid_prj']
        cell = torch.where(mask_n != 0, cell, cell_previous)
        hid = torch.where(mask_n != 0, hid, hid_previous)
        return [cell, hid]

    init = [torch.zeros(B, N, dtype=x.dtype), torch.zeros(B, F, dtype=x.dtype)]
    _, hid_out = _scan(step, [inp, m], init, backwards)
    hid_out = hid_out.transpose(0, 1)
    return hid_out.flip(1) if backwards else hid_out


def hyper_layer(P, x, mask, N, H, Z, clip=0, backwards=False):
    """libs/lasagne_libs/hyper_lhuc_layers.py:216-237,363-507."""
    W_h = _cat(P, 'W_h_{}', G_L); W_x = _cat(P, 'W_x_{}', G_L)
    W_hhat = _cat(P, 'W_hhat_{}', G_L); W_xhat = _cat(P, 'W_xhat_{}', G_L); bhat = _cat(P, 'bhat_{}', G_L, 0)
    W_hhat_h = _cat(P, 'W_hhat_h_{}', G_L); W_hhat_x = _cat(P, 'W_hhat_x_{}', G_L)
    W_hhat_b = _cat(P, 'W_hhat_b_{}', G_L)
    b_hhat_h = _cat(P, 'b_hhat_h_{}', G_L, 0); b_hhat_x = _cat(P, 'b_hhat_x_{}', G_L, 0)
    orig = x.transpose(0, 1)
    inp = orig @ W_x
    m = mask.transpose(0, 1).unsqueeze(-1)
    B = orig.shape[1]

    def step(orig_input_n, input_n, mask_n, hc_prev, hh_prev, c_prev, h_prev):
        hin = torch.cat([h_prev, orig_input_n], 1) @ W_xhat + bhat
        hg = grad_clip(hin + hh_prev @ W_hhat, clip)
        hig = torch.sigmoid(hg[:, 0:H]); hfg = torch.sigmoid(hg[:, H:2 * H])
        hci = torch.tanh(hg[:, 2 * H:3 * H])
        hc = hfg * hc_prev + hig * hci
        hog = torch.sigmoid(hg[:, 3 * H:4 * H])
        hh = hog * torch.tanh(hc)
        z_h = hh @ W_hhat_h + b_hhat_h; z_x = hh @ W_hhat_x + b_hhat_x; z_b = hh @ W_hhat_b
        d_h = torch.cat([z_h[:, k * Z:(k + 1) * Z] @ P['W_hz_' + g] for k, g in enumerate(G_L)], 1)
        d_x = torch.cat([z_x[:, k * Z:(k + 1) * Z] @ P['W_xz_' + g] for k, g in enumerate(G_L)], 1)
        bb = torch.cat([z_b[:, k * Z:(k + 1) * Z] @ P['W_bz_' + g] + P['b_0_' + g]
                        for k, g in enumerate(G_L)], 1)
        inn = input_n * d_x + bb
        gates = grad_clip((h_prev @ W_h) * d_h + inn, clip)
        ig = torch.sigmoid(gates[:, 0:N]); fg = torch.sigmoid(gates[:, N:2 * N])
        ci = torch.tanh(gates[:, 2 * N:3 * N])
        cell = fg * c_prev + ig * ci
        og = torch.sigmoid(gates[:, 3 * N:4 * N])
        hid = og * torch.tanh(cell)
        k = mask_n != 0
        return [torch.where(k, hc, hc_prev), torch.where(k, hh, hh_prev),
                torch.where(k, cell, c_prev), torch.where(k, hid, h_prev)]

    z = lambda n: torch.zeros(B, n, dtype=x.dtype)
    outs = _scan(step, [orig, inp, m], [z(H), z(H), z(N), z(N)], backwards)
    hid_out = outs[3].transpose(0, 1)
    return hid_out.flip(1) if backwards else hid_out


# ------------------------------------------------------------------------------------------------
# The 20 LSTM layer classes of libs/lasagne_libs/recurrent_layers.py, written class by class after
# their get_output_for bodies (NOT from the oracle's variant table, so that it pins that table).
def gate_ln(x, alpha, beta=None):                  # :17-25 (static) and :34-42 (per sample)
    B = x.shape[0]
    ga = alpha.reshape(1 if alpha.dim() == 1 else B, 4, -1)
    gi = x.reshape(B, 4, -1)
    go = (gi - gi.mean(2, keepdim=True)) / torch.sqrt(gi.var(2, keepdim=True, unbiased=False) + EPS)
    go = ga * go
    if beta is not None:
        go = go + beta.reshape(1 if beta.dim() == 1 else B, 4, -1)
    return go.reshape(B, -1)


def _pool_mean_tanh(src, mask, P):                 # :944-946 / :1209-1211
    li = torch.tanh(src @ P['W_cond_prj'] + P['b_cond_prj'])
    return (li * mask.unsqueeze(-1)).sum(1) / mask.sum(1, keepdim=True)


class _TheanoMax(torch.autograd.Function):
    """T.max(x, axis=1) with Theano's gradient (MaxAndArgmax.grad: g * eq(max, x), i.e. EVERY tied
    position receives the gradient; torch.max would pick one).  Ties are the normal case for the
    external i-vector input, which is constant over the frames of an utterance."""
    @staticmethod
    def forward(ctx, x):
        mx = x.max(dim=1).values
        ctx.save_for_backward((x == mx.unsqueeze(1)).to(x.dtype))
        return mx

    @staticmethod
    def backward(ctx, g):
        (eq,) = ctx.saved_tensors
        return g.unsqueeze(1) * eq


def _pool_max_relu(src, mask, P):                  # :1468-1470 / :1733-1735 (mask unused there)
    li = torch.relu(src @ P['W_cond_prj'] + P['b_cond_prj'])
    return _TheanoMax.apply(li)


def rec_layer(P, cls, x, mask, N, clip=0, backwards=False, cond=None, peepholes=False, learn_init=False,
              only_return_final=False, only_return_hidden=True):
    W_in = _cat(P, 'W_in_to_{}', G_R); W_hid = _cat(P, 'W_hid_to_{}', G_R)
    K = W_hid.shape[0]
    inp = x.transpose(0, 1)
    m = mask.transpose(0, 1).unsqueeze(-1)
    B = inp.shape[1]
    has_b = ('b_ingate' in P)
    b_stacked = _cat(P, 'b_{}', G_R, 0) if has_b else None
    alpha_in = alpha_hid = beta = None
    # ---- prologue
    if cls in ('LSTMLayer', 'ProjectLSTMLayer', 'ProjectLSTM_v0_1_Layer', 'RegularLSTM_v0_1_Layer'):
        inp = inp @ W_in + b_stacked                                   # :174, :431, :2007, :3351
    elif cls in ('CondLayerNormProjectLSTMLayer', 'InputCondLayerNormProjectLSTMLayer',
                 'MaxCondLayerNormProjectLSTMLayer', 'MaxInputCondLayerNormProjectLSTMLayer'):
        src = x if cls in ('CondLayerNormProjectLSTMLayer', 'MaxCondLayerNormProjectLSTMLayer') else cond
        li = (_pool_max_relu if cls.startswith('Max') else _pool_mean_tanh)(src, mask, P)
        alpha_in = li @ P['W_alpha_in'] + P['b_alpha_in']
        alpha_hid = li @ P['W_alpha_hid'] + P['b_alpha_hid']
        beta = li @ P['W_beta'] + P['b_beta']
        inp = inp @ W_in + beta                                        # :968, :1233, :1492, :1757
    elif cls in ('CondLayerNormProjectLSTM_v0_1_Layer', 'CondLayerNormProjectLSTM_v0_2_Layer',
                 'CondLayerNormRegularLSTM_v0_1_Layer', 'CondLayerNormRegularLSTM_v0_2_Layer'):
        li = _pool_mean_tanh(x, mask, P)                               # :2799-2801
        alpha_in = li @ P['W_alpha_in'] + P['b_alpha_in']
        alpha_hid = li @ P['W_alpha_hid'] + P['b_alpha_hid']
        beta = li @ P['W_beta'] + b_stacked                            # :2807
        inp = inp @ W_in
    elif cls in ('CondLayerNormLSTMLayer', 'InputCondLayerNormLSTMLayer'):
        li = _pool_mean_tanh(x if cls == 'CondLayerNormLSTMLayer' else cond, mask, P)   # :4902-4904, :5163-5165
        alpha_in = li @ P['W_alpha_in'] + P['b_alpha_in']
        alpha_hid = li @ P['W_alpha_hid'] + P['b_alpha_hid']
        beta = li @ P['W_beta'] + P['b_beta']
        inp = inp @ W_in                                               # :4926
    else:
        inp = inp @ W_in                                               # static LN classes: :701, :2266

    def step(input_n, mask_n, cell_previous, hid_previous):
        hw = hid_previous @ W_hid
        if cls in ('LSTMLayer', 'ProjectLSTMLayer', 'ProjectLSTM_v0_1_Layer', 'RegularLSTM_v0_1_Layer'):
            gates = input_n + hw
        elif cls in ('LayerNormProjectLSTMLayer', 'LayerNormLSTMLayer'):
            gates = ln(input_n, P['W_alpha_in'], P['W_beta_in']) + ln(hw, P['W_alpha_hid'], P['W_beta_hid'])
            gates = gates + b_stacked                                  # :711-713, :4675-4677
        elif cls in ('LayerNormProjectLSTM_v0_1_Layer', 'LayerNormRegularLSTM_v0_1_Layer'):
            gates = ln(input_n, P['W_alpha_in']) + ln(hw, P['W_alpha_hid']) + b_stacked   # :2276-2278
        elif cls in ('LayerNormProjectLSTM_v0_2_Layer', 'LayerNormRegularLSTM_v0_2_Layer'):
            gates = gate_ln(input_n, P['W_alpha_in']) + gate_ln(hw, P['W_alpha_hid']) + b_stacked   # :2531-2533
        elif cls in ('CondLayerNormProjectLSTM_v0_1_Layer', 'CondLayerNormRegularLSTM_v0_1_Layer',
                     'CondLayerNormLSTMLayer', 'InputCondLayerNormLSTMLayer'):
            gates = ln(input_n, alpha_in) + ln(hw, alpha_hid) + beta   # :2822-2824, :4935-4937
        elif cls in ('CondLayerNormProjectLSTM_v0_2_Layer', 'CondLayerNormRegularLSTM_v0_2_Layer'):
            gates = gate_ln(input_n, alpha_in) + gate_ln(hw, alpha_hid) + beta          # :3113-3115
        else:                                                          # sample_ln, beta already in input_n
            gates = ln(input_n, alpha_in) + ln(hw, alpha_hid)          # :976-977, :1241-1242
        ig, fg, ci, og = [gates[:, k * N:(k + 1) * N] for k in range(4)]
        if peepholes:                                                  # :190-192
            ig = ig + cell_previous * P['W_cell_to_ingate']
            fg = fg + cell_previous * P['W_cell_to_forgetgate']
        ig, fg, ci = grad_clip(ig, clip), grad_clip(fg, clip), grad_clip(ci, clip)
        ig = torch.sigmoid(ig); fg = torch.sigmoid(fg); ci = torch.tanh(ci)
        cell = fg * cell_previous + ig * ci
        if peepholes:                                                  # :212-213
            og = og + cell * P['W_cell_to_outgate']
        og = grad_clip(og, clip)
        if 'W_alpha_cell' in P:                                        # :740, :2305, :2849
            outcell = ln(cell, P['W_alpha_cell'], P['num_factors'] if 'num_factors' in P else P['W_beta_cell'])
        else:
            outcell = cell
        outcell = grad_clip(outcell, clip)
        og = torch.sigmoid(og)
        hid = og * torch.tanh(outcell)
        if 'W_hid_prj' in P:
            hid = hid @ P['W_hid_prj']
        cell = torch.where(mask_n != 0, cell, cell_previous)
        hid = torch.where(mask_n != 0, hid, hid_previous)
        return [cell, hid]

    if learn_init:                                                     # :246-248
        ones = torch.ones(B, 1, dtype=x.dtype)
        init = [ones @ P['cell_init'], ones @ P['hid_init']]
    else:
        init = [torch.zeros(B, N, dtype=x.dtype), torch.zeros(B, K, dtype=x.dtype)]
    cell_out, hid_out = _scan(step, [inp, m], init, backwards)

    def shape_out(seq):                                                # :265-273 (and :278-286 for the cells)
        if only_return_final:
            return seq[-1]
        seq = seq.transpose(0, 1)
        return seq.flip(1) if backwards else seq
    hid_out = shape_out(hid_out)
    if only_return_hidden:
        return hid_out
    return torch.cat([hid_out, shape_out(cell_out)], dim=-1)           # :288


def seqsum_lhuc_net(ss, layers, x, mask, N, npl, clip=0):
    """build_deep_lhuclstm_seqsum wiring (libs/deep_lhuclstm_builder.py:8-45, libs/builder_utils.py:48-62)
    for unidirectional layers: ss = [W0, b0, ..., W_out, b_out]; layers = list of LHUC layer dicts."""
    nl = len(ss) // 2 - 1
    h = x.reshape(-1, x.shape[-1])
    for i in range(nl):
        h = torch.tanh(h @ ss[2 * i] + ss[2 * i + 1])
    h = h @ ss[2 * nl] + ss[2 * nl + 1]
    emb = h.reshape(x.shape[0], x.shape[1], -1).mean(dim=1)
    out = x
    for P in layers:
        hid = lstm_layer(P, out, mask, N, clip=clip)
        pred = emb
        for i in range(npl):
            pred = torch.relu(pred @ P['W_e_h_%d' % i] + P['b_e_h_%d' % i])
        sf = pred @ P['W_e_h_%d' % npl] + P['b_e_h_%d' % npl]
        out = hid * torch.relu(sf).unsqueeze(1)
    return out


def scaling_hyper_layer(P, x, mask, N, H, clip=0, backwards=False, use_peepholes=False, use_layer_norm=False,
                        learn_init=False, return_inner=False, equal_scale=False):
    """libs/lasagne_libs/hyper_layers.py:850-1205, line by line.  Without use_peepholes the peephole vectors and the
    scale_outer_cell map are zero constants (:585,:716-723); the cell gate's peephole is a zero constant always
    (cell_trainable=False, :642-644,:791-793).  Returns outer hid (or [inner hid, outer hid]).
    equal_scale=True: EqualScaleHyperLSTMLayer (:1537-1890), the same step with an inner cell of N columns per gate
    (slice_outer, :1660-1689) and inner_hid projected by W_inner_hid_prj (:1709)."""
    HR = H                                  # width of the carried inner hidden state
    if equal_scale:
        H = N                               # width of the inner cell
    c = lambda fmt, dim=1: _cat(P, fmt, G_R, dim)
    W_ii = c('W_inner_in_to_inner_{}'); W_hi = c('W_inner_hid_to_inner_{}'); W_oi = c('W_outer_hid_to_inner_{}')
    b_i = c('b_inner_{}', 0)
    W_in = c('W_inner_hid_to_outer_in_{}'); b_in = c('b_inner_hid_to_outer_in_{}', 0)
    W_hd = c('W_inner_hid_to_outer_hid_{}'); b_hd = c('b_inner_hid_to_outer_hid_{}', 0)
    W_bs = c('W_inner_hid_to_outer_bias_{}'); b_bs = c('b_inner_hid_to_outer_bias_{}', 0)
    W_oo_in = c('W_outer_in_to_outer_{}'); W_oo = c('W_outer_hid_to_outer_{}')
    zH = torch.zeros(H, dtype=x.dtype); zHN = torch.zeros(HR, N, dtype=x.dtype); zN = torch.zeros(N, dtype=x.dtype)
    pk = lambda g: P['W_inner_cell_to_inner_' + g] if (use_peepholes and g != 'cell') else zH
    W_cl = torch.cat([P['W_inner_hid_to_outer_cell_' + g] if (use_peepholes and g != 'cell') else zHN for g in G_R], 1)
    b_cl = torch.cat([P['b_inner_hid_to_outer_cell_' + g] if (use_peepholes and g != 'cell') else zN for g in G_R], 0)

    def lnorm(v, pre, g):
        if not use_layer_norm:
            return v
        out = (v - v.mean(1, keepdim=True)) / torch.sqrt(v.var(1, unbiased=False, keepdim=True) + EPS)
        return P['W_%s_ln_%s' % (pre, g)][None, :] * out + P['b_%s_ln_%s' % (pre, g)][None, :]

    inp = x.transpose(0, 1)
    m = mask.transpose(0, 1).unsqueeze(-1)
    B = inp.shape[1]
    inner_input = inp @ W_ii + b_i
    outer_input = inp @ W_oo_in
    sl = lambda v, n, U: v[:, n * U:(n + 1) * U]

    def step(inner_input_n, outer_input_n, mask_n, ic_prev, ih_prev, oc_prev, oh_prev):
        ig = inner_input_n + ih_prev @ W_hi + oh_prev @ W_oi                                   # :973-975
        i_in = sl(ig, 0, H) + ic_prev * pk('ingate'); i_f = sl(ig, 1, H) + ic_prev * pk('forgetgate')
        i_c = sl(ig, 2, H) + ic_prev * pk('cell')                                              # :977-979
        i_in = grad_clip(lnorm(i_in, 'inner', 'ingate'), clip); i_f = grad_clip(lnorm(i_f, 'inner', 'forgetgate'), clip)
        i_c = grad_clip(lnorm(i_c, 'inner', 'cell'), clip)                                     # :981-1001
        inner_cell = torch.sigmoid(i_f) * ic_prev + torch.sigmoid(i_in) * torch.tanh(i_c)      # :1003-1005
        i_o = sl(ig, 3, H) + inner_cell * pk('outgate')                                        # :1006
        i_o = grad_clip(lnorm(i_o, 'inner', 'outgate'), clip)
        inner_outcell = grad_clip(lnorm(inner_cell, 'inner', 'outcell'), clip)                 # :1007-1022
        inner_hid = torch.sigmoid(i_o) * torch.tanh(inner_outcell)                             # :1025
        if equal_scale:
            inner_hid = inner_hid @ P['W_inner_hid_prj']                                       # :1709
        s_in = inner_hid @ W_in + b_in; s_hid = inner_hid @ W_hd + b_hd                        # :1030-1033
        s_cl = inner_hid @ W_cl + b_cl; s_b = inner_hid @ W_bs + b_bs
        og = s_in * outer_input_n + s_hid * (oh_prev @ W_oo) + s_b                             # :1038-1040
        o_in = sl(og, 0, N) + oc_prev * sl(s_cl, 0, N); o_f = sl(og, 1, N) + oc_prev * sl(s_cl, 1, N)
        o_c = sl(og, 2, N) + oc_prev * sl(s_cl, 2, N)                                          # :1042-1044
        o_in = grad_clip(lnorm(o_in, 'outer', 'ingate'), clip); o_f = grad_clip(lnorm(o_f, 'outer', 'forgetgate'), clip)
        o_c = grad_clip(lnorm(o_c, 'outer', 'cell'), clip)
        outer_cell = torch.sigmoid(o_f) * oc_prev + torch.sigmoid(o_in) * torch.tanh(o_c)      # :1069-1071
        o_o = sl(og, 3, N) + outer_cell * sl(s_cl, 3, N)                                       # :1072
        o_o = grad_clip(lnorm(o_o, 'outer', 'outgate'), clip)
        outer_outcell = grad_clip(lnorm(outer_cell, 'outer', 'outcell'), clip)
        outer_hid = torch.sigmoid(o_o) * torch.tanh(outer_outcell)                             # :1091
        k = mask_n != 0
        return [torch.where(k, inner_cell, ic_prev), torch.where(k, inner_hid, ih_prev),
                torch.where(k, outer_cell, oc_prev), torch.where(k, outer_hid, oh_prev)]

    ones = torch.ones(B, 1, dtype=x.dtype)
    init = lambda nm, n: ones @ P[nm] if learn_init else torch.zeros(B, n, dtype=x.dtype)     # :1127-1131
    outs = _scan(step, [inner_input, outer_input, m],
                 [init('inner_cell_init', H), init('inner_hid_init', HR), init('outer_cell_init', N), init('outer_hid_init', N)],
                 backwards)
    hid = outs[3].transpose(0, 1)
    hid = hid.flip(1) if backwards else hid
    if return_inner:
        ih = outs[1].transpose(0, 1)
        return [ih.flip(1) if backwards else ih, hid]
    return hid


def hyperlhuc_layer(P, x, mask, N, H, Z, clip=0, backwards=False, reparam='relu'):
    """libs/lasagne_libs/hyper_lhuc_layers.py:580-678 (step, get_output_for), compute_eq10 :216-237,
    reparam_* :17-27."""
    W_h = _cat(P, 'W_h_{}', G_L); W_x = _cat(P, 'W_x_{}', G_L); b = _cat(P, 'b_{}', G_L, 0)
    W_hhat = _cat(P, 'W_hhat_{}', G_L); W_xhat = _cat(P, 'W_xhat_{}', G_L); bhat = _cat(P, 'bhat_{}', G_L, 0)
    rp = dict(relu=torch.relu, exp=torch.exp, identity=lambda v: v,
              **{'2sigmoid': lambda v: 2 / (1 + torch.exp(-v))})[reparam]
    orig = x.transpose(0, 1)
    inp = orig @ W_x + b
    m = mask.transpose(0, 1).unsqueeze(-1)
    B = orig.shape[1]

    def step(orig_input_n, input_n, mask_n, hc_prev, hh_prev, c_prev, h_prev):
        hin = torch.cat([h_prev, orig_input_n], 1) @ W_xhat + bhat
        hg = grad_clip(hin + hh_prev @ W_hhat, clip)
        hig = torch.sigmoid(hg[:, 0:H]); hfg = torch.sigmoid(hg[:, H:2 * H])
        hci = torch.tanh(hg[:, 2 * H:3 * H])
        hc = hfg * hc_prev + hig * hci
        hog = torch.sigmoid(hg[:, 3 * H:4 * H])
        hh = hog * torch.tanh(hc)
        z_h = hh @ P['W_hhat_h'] + P['b_hhat_h']
        d_h = z_h @ P['W_hz']
        gates = grad_clip(h_prev @ W_h + input_n, clip)
        ig = torch.sigmoid(gates[:, 0:N]); fg = torch.sigmoid(gates[:, N:2 * N])
        ci = torch.tanh(gates[:, 2 * N:3 * N])
        cell = fg * c_prev + ig * ci
        og = torch.sigmoid(gates[:, 3 * N:4 * N])
        hid = og * torch.tanh(cell)
        hid = hid * rp(d_h)
        k = mask_n != 0
        return [torch.where(k, hc, hc_prev), torch.where(k, hh, hh_prev),
                torch.where(k, cell, c_prev), torch.where(k, hid, h_prev)]

    z = lambda n: torch.zeros(B, n, dtype=x.dtype)
    outs = _scan(step, [orig, inp, m], [z(H), z(H), z(N), z(N)], backwards)
    hid_out = outs[3].transpose(0, 1)
    return hid_out.flip(1) if backwards else hid_out


def prjhyper_layer(P, x, mask, N, F, clip=0, backwards=False):
    """libs/lasagne_libs/hyper_layers.py:238-519 (get_output_for of ProjectionHyperLSTMLayer); ln :9-12."""
    c = lambda fmt, dim=1: _cat(P, fmt, G_R, dim)
    W_in_to_inner = c('W_in_to_inner_{}'); W_inner_to_inner = c('W_inner_to_inner_{}')
    W_outer_to_inner = c('W_outer_to_inner_{}'); b_inner = c('b_inner_{}', 0)
    W_in_to_outer = c('W_in_to_outer_{}'); W_outer_to_outer = c('W_outer_to_outer_{}')
    W_inner_to_outer = torch.cat([c('W_inner_to_in_{}'), c('W_inner_to_outer_{}'), c('W_inner_to_bias_{}')], 1)
    b_inner_to_outer = torch.cat([c('b_inner_to_in_{}', 0), c('b_inner_to_outer_{}', 0), c('b_inner_to_bias_{}', 0)], 0)
    eps = float(torch.finfo(torch.float32).eps)

    def ln(v, alpha, beta):
        out = (v - v.mean(1, keepdim=True)) / torch.sqrt(v.var(1, unbiased=False, keepdim=True) + eps)
        return alpha[None, :] * out + beta[None, :]

    inp = x.transpose(0, 1)
    m = mask.transpose(0, 1).unsqueeze(-1)
    B = inp.shape[1]
    inner_input = inp @ W_in_to_inner + b_inner
    outer_input = inp @ W_in_to_outer
    su = lambda v, n: v[:, n * N:(n + 1) * N]
    sb = lambda v, n: v[:, 4 * n * N:4 * (n + 1) * N]

    def step(inner_input_n, outer_input_n, mask_n, ic_prev, ih_prev, oc_prev, oh_prev):
        ig = inner_input_n + ih_prev @ W_inner_to_inner + oh_prev @ W_outer_to_inner
        i_in = torch.sigmoid(grad_clip(su(ig, 0), clip)); i_f = torch.sigmoid(grad_clip(su(ig, 1), clip))
        i_c = grad_clip(su(ig, 2), clip); i_o = torch.sigmoid(grad_clip(su(ig, 3), clip))
        inner_cell = i_f * ic_prev + i_in * torch.tanh(i_c)
        inner_outcell = grad_clip(ln(inner_cell, P['W_inner_cell_ln'], P['b_inner_cell_ln']), clip)
        inner_hid = (i_o * torch.tanh(inner_outcell)) @ P['W_inner_prj']
        scale_outer = inner_hid @ W_inner_to_outer + b_inner_to_outer
        og = outer_input_n * sb(scale_outer, 0) + (oh_prev @ W_outer_to_outer) * sb(scale_outer, 1) + sb(scale_outer, 2)
        o_in = torch.sigmoid(grad_clip(su(og, 0), clip)); o_f = torch.sigmoid(grad_clip(su(og, 1), clip))
        o_c = grad_clip(su(og, 2), clip); o_o = torch.sigmoid(grad_clip(su(og, 3), clip))
        outer_cell = o_f * oc_prev + o_in * torch.tanh(o_c)
        outer_outcell = grad_clip(ln(outer_cell, P['W_outer_cell_ln'], P['b_outer_cell_ln']), clip)
        outer_hid = (o_o * torch.tanh(outer_outcell)) @ P['W_outer_prj']
        k = mask_n != 0
        return [torch.where(k, inner_cell, ic_prev), torch.where(k, inner_hid, ih_prev),
                torch.where(k, outer_cell, oc_prev), torch.where(k, outer_hid, oh_prev)]

    z = lambda n: torch.zeros(B, n, dtype=x.dtype)
    outs = _scan(step, [inner_input, outer_input, m], [z(N), z(F), z(N), z(F)], backwards)
    hi = outs[1].transpose(0, 1); ho = outs[3].transpose(0, 1)
    if backwards:
        hi = hi.flip(1); ho = ho.flip(1)
    return torch.cat([hi, ho], -1)

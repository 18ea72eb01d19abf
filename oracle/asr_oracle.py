"""CPU oracle for the acoustic-model hot path of gunkisu/asr  --  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement (float64 by default) of the reference's Theano/Lasagne step
functions, their BPTT, the framewise softmax-CE loss and the Adam update.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it; the product
package `asr_amd` never does.

PARITY UNPINNED: the reference ships no golden vectors, fixtures or known-answer tests for this
path and Theano/Lasagne are not importable in this image, so the oracle
cannot be checked against outputs of the reference itself.  It is instead pinned by (1) a line-by
-line torch float64 transliteration of the reference step functions differentiated with torch
autograd (tests/test_oracle_autograd.py) and (2) central finite differences.

Every function cites the reference file:line it restates (paths relative to /root/reference).

Third-party semantics restated here (Theano / Lasagne, not vendored in the reference, no version
pin exists):  `theano.gradient.grad_clip(x, -c, c)` = identity forward, elementwise clip of the
incoming gradient backward, skipped when c == 0;  `T.switch(m, a, b)` routes gradient to `a` where
m != 0 else to `b`;  `T.var` = population variance;  `theano.scan(go_backwards=True)` iterates
t = T-1..0 over the padded array;  Lasagne `DenseLayer` = x.W + b, W (in, out).
"""
import numpy as np

EPS = float(np.finfo(np.float32).eps)   # libs/lhuclstm_layers.py:13-14 (floatX = float32)


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


def _clip(g, c):
    return np.clip(g, -c, c) if c else g


# --------------------------------------------------------------------------------------------
# layer norm  (libs/lhuclstm_layers.py:16-21, libs/lasagne_libs/recurrent_layers.py:10-15,27-32)
# --------------------------------------------------------------------------------------------
def ln_fwd(x, alpha, beta=None, groups=1):
    """y = alpha * (x - mean) / sqrt(var + eps) [+ beta]; alpha/beta are (n,) or per-sample (B,n).
    groups=4: gate_ln / sample_gate_ln (recurrent_layers.py:17-25,34-42): each of the 4 contiguous
    gate blocks of a row is normalised separately."""
    B, n = x.shape
    xg = x.reshape(B, groups, n // groups)
    mu = xg.mean(axis=2, keepdims=True)
    var = xg.var(axis=2, keepdims=True)
    s = np.sqrt(var + EPS)
    xh = ((xg - mu) / s).reshape(B, n)
    y = alpha * xh
    if beta is not None:
        y = y + beta
    return y, (xh, s, groups)


def ln_bwd(dy, alpha, cache):
    xh, s, groups = cache
    B, n = dy.shape
    dxh = (dy * alpha).reshape(B, groups, n // groups)
    xg = xh.reshape(B, groups, n // groups)
    dx = (dxh - dxh.mean(axis=2, keepdims=True) - xg * (dxh * xg).mean(axis=2, keepdims=True)) / s
    return dx.reshape(B, n), dy * xh, dy        # dx, per-sample dalpha, per-sample dbeta


# --------------------------------------------------------------------------------------------
# LSTM family scan.  One parameterised restatement of
#   libs/lhuclstm_layers.py:73-130,172-218      (LSTMLayer: optional proj, LN alpha-only + cell LN)
#   libs/lasagne_libs/recurrent_layers.py:437-498   (ProjectLSTMLayer)
#   libs/lasagne_libs/recurrent_layers.py:707-763   (LayerNormProjectLSTMLayer)
#   libs/lasagne_libs/recurrent_layers.py:1237-1289 (InputCondLayerNormProjectLSTMLayer)
# spec keys: N, P (0 = no projection), ln ('none'|'static'|'sample'), cell_ln, grad_clip, backwards
# p keys: W_h (P|N,4N), b (4N)|None (added in the step after LN), W_p (N,P)|None,
#         alpha_in, beta_in|None, alpha_hid, beta_hid|None  ((4N,) static or (B,4N) per sample),
#         alpha_c, beta_c (N,)
# xin (T,B,4N) is whatever the reference feeds the scan as `input_n` (x.W_x, + b or + beta where the
# reference adds it before the scan).
# --------------------------------------------------------------------------------------------
def lstm_scan_fwd(xin, mask, p, spec, c0=None, h0=None):
    T, B, _ = xin.shape
    N = spec['N']
    P = spec['P'] if spec['P'] else N
    ln = spec.get('ln', 'none')
    dt = xin.dtype
    c = np.zeros((B, N), dt) if c0 is None else c0.astype(dt)
    h = np.zeros((B, P), dt) if h0 is None else h0.astype(dt)
    order = range(T - 1, -1, -1) if spec.get('backwards') else range(T)
    hid = np.zeros((T, B, P), dt)
    cell = np.zeros((T, B, N), dt)
    cache = [None] * T
    for t in order:
        m = mask[t][:, None] != 0
        a = h @ p['W_h']
        ci = ch = cc = None
        if ln != 'none':
            grp = 4 if spec.get('per_gate') else 1
            gi, ci = ln_fwd(xin[t], p['alpha_in'], p.get('beta_in'), grp)
            gh, ch = ln_fwd(a, p['alpha_hid'], p.get('beta_hid'), grp)
            gates = gi + gh
        else:
            gates = xin[t] + a
        if p.get('b') is not None:
            gates = gates + p['b']
        pe = p.get('peep')            # (w_ci, w_cf, w_co): recurrent_layers.py:190-192 (previous cell), :212-213 (new cell)
        gi_, gf_ = gates[:, 0 * N:1 * N], gates[:, 1 * N:2 * N]
        if pe is not None:
            gi_ = gi_ + c * pe[0]; gf_ = gf_ + c * pe[1]
        i = _sig(gi_)
        f = _sig(gf_)
        g = np.tanh(gates[:, 2 * N:3 * N])
        c_new = f * c + i * g
        go_ = gates[:, 3 * N:4 * N]
        if pe is not None:
            go_ = go_ + c_new * pe[2]
        o = _sig(go_)
        if spec.get('cell_ln'):
            outcell, cc = ln_fwd(c_new, p['alpha_c'], p['beta_c'])
        else:
            outcell = c_new
        tc = np.tanh(outcell)
        r = o * tc
        h_new = r @ p['W_p'] if p.get('W_p') is not None else r
        cache[t] = (m, c, h, i, f, g, o, tc, r, ci, ch, cc)
        c = np.where(m, c_new, c)
        h = np.where(m, h_new, h)
        hid[t] = h
        cell[t] = c
    return hid, cell, cache


def carried_r(cache, spec, r0=None):
    """(T,B,N): the masked pre-projection state r_t = o*tanh(outcell) that lstm_scan_fwd carried (its own
    state is h = r.W_p; the HIP scan carries r and folds W_p into the recurrent matrix)."""
    T = len(cache)
    order = range(T - 1, -1, -1) if spec.get('backwards') else range(T)
    r = None
    out = [None] * T
    for t in order:
        m, rn = cache[t][0], cache[t][8]
        if r is None:
            r = np.zeros_like(rn) if r0 is None else r0
        r = np.where(m, rn, r)
        out[t] = r
    return np.stack(out)


def lstm_scan_bwd(dhid, cache, p, spec, dcell=None):
    """BPTT through lstm_scan_fwd.  dhid (T,B,P) = dLoss/d hid_out.  Returns dxin, grads, (dc0, dh0)."""
    T = dhid.shape[0]
    N = spec['N']
    clipv = spec.get('grad_clip', 0)
    ln = spec.get('ln', 'none')
    dt = dhid.dtype
    B = dhid.shape[1]
    order = list(range(T - 1, -1, -1) if spec.get('backwards') else range(T))
    g_ = {k: np.zeros_like(v) for k, v in p.items() if v is not None and k != 'peep'}
    pe = p.get('peep')
    if pe is not None:
        g_['peep'] = [np.zeros_like(w) for w in pe]
    dh = np.zeros_like(dhid[0])
    dc = np.zeros((B, N), dt)
    dxin = np.zeros((T, B, 4 * N), dt)
    for t in reversed(order):
        (m, c_prev, h_prev, i, f, g, o, tc, r, ci, ch, cc) = cache[t]
        dh_t = dhid[t] + dh
        dc_t = dc + (dcell[t] if dcell is not None else 0)
        dh_new = np.where(m, dh_t, 0)
        dc_new = np.where(m, dc_t, 0)
        dh_prev = np.where(m, 0, dh_t)
        dc_prev = np.where(m, 0, dc_t)
        if p.get('W_p') is not None:
            g_['W_p'] += r.T @ dh_new
            dr = dh_new @ p['W_p'].T
        else:
            dr = dh_new
        do = dr * tc
        dout = _clip(dr * o * (1 - tc * tc), clipv)
        if spec.get('cell_ln'):
            dcn, dal, dbe = ln_bwd(dout, p['alpha_c'], cc)
            g_['alpha_c'] += dal.sum(0)
            g_['beta_c'] += dbe.sum(0)
            dc_new = dc_new + dcn
        else:
            dc_new = dc_new + dout
        dgo = _clip(do * o * (1 - o), clipv)
        if pe is not None:                 # the out gate looks at the NEW cell (:212-213)
            dc_new = dc_new + dgo * pe[2]
        dgates = np.concatenate([_clip(dc_new * g * i * (1 - i), clipv), _clip(dc_new * c_prev * f * (1 - f), clipv),
                                 _clip(dc_new * i * (1 - g * g), clipv), dgo], axis=1)
        dc_prev = dc_prev + dc_new * f
        if pe is not None:
            dgi, dgf = dgates[:, 0 * N:1 * N], dgates[:, 1 * N:2 * N]
            dc_prev = dc_prev + dgi * pe[0] + dgf * pe[1]
            g_['peep'][0] += (dgi * c_prev).sum(0); g_['peep'][1] += (dgf * c_prev).sum(0)
            g_['peep'][2] += (dgo * (f * c_prev + i * g)).sum(0)
        if p.get('b') is not None:       # (4N,) static bias, or (B,4N) per-sample beta added after LN
            g_['b'] += dgates if p['b'].ndim == 2 else dgates.sum(0)
        if ln != 'none':
            dx_t, dal, dbe = ln_bwd(dgates, p['alpha_in'], ci)
            g_['alpha_in'] += dal if ln == 'sample' else dal.sum(0)
            if p.get('beta_in') is not None:
                g_['beta_in'] += dbe if ln == 'sample' else dbe.sum(0)
            da, dal, dbe = ln_bwd(dgates, p['alpha_hid'], ch)
            g_['alpha_hid'] += dal if ln == 'sample' else dal.sum(0)
            if p.get('beta_hid') is not None:
                g_['beta_hid'] += dbe if ln == 'sample' else dbe.sum(0)
        else:
            dx_t = dgates
            da = dgates
        g_['W_h'] += h_prev.T @ da
        dh_prev = dh_prev + da @ p['W_h'].T
        dxin[t] = dx_t
        dh, dc = dh_prev, dc_prev
    return dxin, g_, (dc, dh)


# --------------------------------------------------------------------------------------------
# HyperLSTM scan  (libs/lasagne_libs/hyper_lhuc_layers.py:216-237 eq10, :363-412 step,
#                  :414-425 step_masked, :427-507 get_output_for)
# p keys: W_xhat (N+D,4H) rows [h ; x] (:218), bhat (4H), W_hhat (H,4H),
#         W_hhat_h, W_hhat_x, W_hhat_b (H,4Z), b_hhat_h, b_hhat_x (4Z),
#         W_hz, W_xz, W_bz (4,Z,N), b_0 (4,N), W_h (N,4N), W_x (D,4N)
# --------------------------------------------------------------------------------------------
def hyper_scan_fwd(x, mask, p, spec):
    T, B, D = x.shape
    N, H, Z = spec['N'], spec['H'], spec['Z']
    dt = x.dtype
    xproj = x @ p['W_x']                                  # :469
    hc = np.zeros((B, H), dt); hh = np.zeros((B, H), dt)
    c = np.zeros((B, N), dt); h = np.zeros((B, N), dt)
    order = range(T - 1, -1, -1) if spec.get('backwards') else range(T)
    hid = np.zeros((T, B, N), dt)
    cache = [None] * T
    for t in order:
        m = mask[t][:, None] != 0
        hx = np.concatenate([h, x[t]], axis=1)            # :218
        hg = hx @ p['W_xhat'] + p['bhat'] + hh @ p['W_hhat']   # :220-221
        hi = _sig(hg[:, 0:H]); hf = _sig(hg[:, H:2 * H])
        hgc = np.tanh(hg[:, 2 * H:3 * H]); ho = _sig(hg[:, 3 * H:4 * H])
        hc_new = hf * hc + hi * hgc                       # :232
        thc = np.tanh(hc_new)
        hh_new = ho * thc                                 # :235
        z_h = hh_new @ p['W_hhat_h'] + p['b_hhat_h']      # :368
        z_x = hh_new @ p['W_hhat_x'] + p['b_hhat_x']      # :369
        z_b = hh_new @ p['W_hhat_b']                      # :370
        d_h = np.concatenate([z_h[:, k * Z:(k + 1) * Z] @ p['W_hz'][k] for k in range(4)], axis=1)
        d_x = np.concatenate([z_x[:, k * Z:(k + 1) * Z] @ p['W_xz'][k] for k in range(4)], axis=1)
        bb = np.concatenate([z_b[:, k * Z:(k + 1) * Z] @ p['W_bz'][k] + p['b_0'][k]
                             for k in range(4)], axis=1)  # :380-388
        a = h @ p['W_h']                                  # :394
        gates = a * d_h + (xproj[t] * d_x + bb)           # :393-395
        i = _sig(gates[:, 0:N]); f = _sig(gates[:, N:2 * N])
        g = np.tanh(gates[:, 2 * N:3 * N]); o = _sig(gates[:, 3 * N:4 * N])
        c_new = f * c + i * g                             # :408
        tc = np.tanh(c_new)
        h_new = o * tc                                    # :411
        cache[t] = (m, hc, hh, c, h, hx, hi, hf, hgc, ho, thc, hh_new, z_h, z_x, z_b,
                    d_h, d_x, a, i, f, g, o, tc)
        hc = np.where(m, hc_new, hc); hh = np.where(m, hh_new, hh)     # :419-423
        c = np.where(m, c_new, c); h = np.where(m, h_new, h)
        hid[t] = h
    return hid, (cache, xproj)


def hyper_scan_bwd(dhid, x, fwd_cache, p, spec):
    cache, xproj = fwd_cache
    T, B, D = x.shape
    N, H, Z = spec['N'], spec['H'], spec['Z']
    clipv = spec.get('grad_clip', 0)
    dt = dhid.dtype
    order = list(range(T - 1, -1, -1) if spec.get('backwards') else range(T))
    g_ = {k: np.zeros_like(v) for k, v in p.items()}
    dh = np.zeros((B, N), dt); dc = np.zeros((B, N), dt)
    dhh = np.zeros((B, H), dt); dhc = np.zeros((B, H), dt)
    dx = np.zeros((T, B, D), dt)
    dxproj = np.zeros((T, B, 4 * N), dt)
    for t in reversed(order):
        (m, hc_prev, hh_prev, c_prev, h_prev, hx, hi, hf, hgc, ho, thc, hh_new, z_h, z_x, z_b,
         d_h, d_x, a, i, f, g, o, tc) = cache[t]
        dh_t = dhid[t] + dh
        dh_new = np.where(m, dh_t, 0); dh_prev = np.where(m, 0, dh_t)
        dc_new = np.where(m, dc, 0); dc_prev = np.where(m, 0, dc)
        dhh_new = np.where(m, dhh, 0); dhh_prev = np.where(m, 0, dhh)
        dhc_new = np.where(m, dhc, 0); dhc_prev = np.where(m, 0, dhc)
        # main cell (:401-411)
        do = dh_new * tc
        dc_new = dc_new + dh_new * o * (1 - tc * tc)
        dgates = np.concatenate([dc_new * g * i * (1 - i), dc_new * c_prev * f * (1 - f),
                                 dc_new * i * (1 - g * g), do * o * (1 - o)], axis=1)
        dgates = _clip(dgates, clipv)                     # :397-399 one clip on the whole vector
        dc_prev = dc_prev + dc_new * f
        da = dgates * d_h
        dd_h = dgates * a
        dxproj[t] = dgates * d_x
        dd_x = dgates * xproj[t]
        dbb = dgates
        g_['W_h'] += h_prev.T @ da
        dh_prev = dh_prev + da @ p['W_h'].T
        dz_h = np.zeros_like(z_h); dz_x = np.zeros_like(z_x); dz_b = np.zeros_like(z_b)
        for k in range(4):
            sl = slice(k * Z, (k + 1) * Z); sn = slice(k * N, (k + 1) * N)
            g_['W_hz'][k] += z_h[:, sl].T @ dd_h[:, sn]
            g_['W_xz'][k] += z_x[:, sl].T @ dd_x[:, sn]
            g_['W_bz'][k] += z_b[:, sl].T @ dbb[:, sn]
            g_['b_0'][k] += dbb[:, sn].sum(0)
            dz_h[:, sl] = dd_h[:, sn] @ p['W_hz'][k].T
            dz_x[:, sl] = dd_x[:, sn] @ p['W_xz'][k].T
            dz_b[:, sl] = dbb[:, sn] @ p['W_bz'][k].T
        g_['W_hhat_h'] += hh_new.T @ dz_h; g_['b_hhat_h'] += dz_h.sum(0)
        g_['W_hhat_x'] += hh_new.T @ dz_x; g_['b_hhat_x'] += dz_x.sum(0)
        g_['W_hhat_b'] += hh_new.T @ dz_b
        dhh_new = dhh_new + dz_h @ p['W_hhat_h'].T + dz_x @ p['W_hhat_x'].T + dz_b @ p['W_hhat_b'].T
        # hyper cell (:226-235)
        dho = dhh_new * thc
        dhc_new = dhc_new + dhh_new * ho * (1 - thc * thc)
        dhg = np.concatenate([dhc_new * hgc * hi * (1 - hi), dhc_new * hc_prev * hf * (1 - hf),
                              dhc_new * hi * (1 - hgc * hgc), dho * ho * (1 - ho)], axis=1)
        dhg = _clip(dhg, clipv)                           # :222-224
        dhc_prev = dhc_prev + dhc_new * hf
        g_['W_xhat'] += hx.T @ dhg
        g_['bhat'] += dhg.sum(0)
        g_['W_hhat'] += hh_prev.T @ dhg
        dhx = dhg @ p['W_xhat'].T
        dh_prev = dh_prev + dhx[:, :N]
        dx[t] = dhx[:, N:]
        dhh_prev = dhh_prev + dhg @ p['W_hhat'].T
        dh, dc, dhh, dhc = dh_prev, dc_prev, dhh_prev, dhc_prev
    g_['W_x'] += x.reshape(T * B, D).T @ dxproj.reshape(T * B, 4 * N)
    dx += dxproj @ p['W_x'].T
    return dx, g_


# --------------------------------------------------------------------------------------------
# HyperLHUCLSTMLayer scan (libs/lasagne_libs/hyper_lhuc_layers.py:509-678): the hyper cell of eq.10
# (shared compute_eq10 :216-237), a PLAIN main LSTM (gates = h.W_h + x.W_x + b, one grad_clip on the
# whole vector :593-597), and LHUC: hid = o*tanh(c) * reparam((hhat.W_hhat_h + b_hhat_h).W_hz), the
# scaled hid being the recurrent state (:585-613).  reparam: relu | exp | identity | 2sigmoid (:17-27).
# p keys: W_xhat (N+D,4H), bhat (4H), W_hhat (H,4H), W_hhat_h (H,Z), b_hhat_h (Z), W_hz (Z,N),
#         W_h (N,4N), W_x (D,4N), b (4N)
# --------------------------------------------------------------------------------------------
def _reparam(kind, v):
    if kind == 'relu':
        return np.maximum(v, 0), (v > 0).astype(v.dtype)
    if kind == 'exp':
        e = np.exp(v); return e, e
    if kind == 'identity':
        return v, np.ones_like(v)
    if kind == '2sigmoid':
        sg = _sig(v); return 2 * sg, 2 * sg * (1 - sg)
    raise KeyError(kind)


def hyperlhuc_scan_fwd(x, mask, p, spec):
    T, B, D = x.shape
    N, H = spec['N'], spec['H']
    dt = x.dtype
    xproj = x @ p['W_x'] + p['b']                         # :645
    hc = np.zeros((B, H), dt); hh = np.zeros((B, H), dt)
    c = np.zeros((B, N), dt); h = np.zeros((B, N), dt)
    order = range(T - 1, -1, -1) if spec.get('backwards') else range(T)
    hid = np.zeros((T, B, N), dt)
    cache = [None] * T
    for t in order:
        m = mask[t][:, None] != 0
        hx = np.concatenate([h, x[t]], axis=1)
        hg = hx @ p['W_xhat'] + p['bhat'] + hh @ p['W_hhat']
        hi = _sig(hg[:, 0:H]); hf = _sig(hg[:, H:2 * H])
        hgc = np.tanh(hg[:, 2 * H:3 * H]); ho = _sig(hg[:, 3 * H:4 * H])
        hc_new = hf * hc + hi * hgc
        thc = np.tanh(hc_new)
        hh_new = ho * thc
        z_h = hh_new @ p['W_hhat_h'] + p['b_hhat_h']      # :585
        d_h = z_h @ p['W_hz']                             # :588
        gates = h @ p['W_h'] + xproj[t]                   # :593
        i = _sig(gates[:, 0:N]); f = _sig(gates[:, N:2 * N])
        g = np.tanh(gates[:, 2 * N:3 * N]); o = _sig(gates[:, 3 * N:4 * N])
        c_new = f * c + i * g                             # :606
        tc = np.tanh(c_new)
        r0 = o * tc                                       # :609
        sc, dsc = _reparam(spec.get('reparam', 'relu'), d_h)
        h_new = r0 * sc                                   # :612
        cache[t] = (m, hc, hh, c, h, hx, hi, hf, hgc, ho, thc, hh_new, z_h, i, f, g, o, tc, r0, sc, dsc)
        hc = np.where(m, hc_new, hc); hh = np.where(m, hh_new, hh)
        c = np.where(m, c_new, c); h = np.where(m, h_new, h)
        hid[t] = h
    return hid, (cache, xproj)


def hyperlhuc_scan_bwd(dhid, x, fwd_cache, p, spec):
    cache, xproj = fwd_cache
    T, B, D = x.shape
    N, H = spec['N'], spec['H']
    clipv = spec.get('grad_clip', 0)
    dt = dhid.dtype
    order = list(range(T - 1, -1, -1) if spec.get('backwards') else range(T))
    g_ = {k: np.zeros_like(v) for k, v in p.items()}
    dh = np.zeros((B, N), dt); dc = np.zeros((B, N), dt)
    dhh = np.zeros((B, H), dt); dhc = np.zeros((B, H), dt)
    dx = np.zeros((T, B, D), dt)
    dxproj = np.zeros((T, B, 4 * N), dt)
    for t in reversed(order):
        (m, hc_prev, hh_prev, c_prev, h_prev, hx, hi, hf, hgc, ho, thc, hh_new, z_h, i, f, g, o, tc,
         r0, sc, dsc) = cache[t]
        dh_t = dhid[t] + dh
        dh_new = np.where(m, dh_t, 0); dh_prev = np.where(m, 0, dh_t)
        dc_new = np.where(m, dc, 0); dc_prev = np.where(m, 0, dc)
        dhh_new = np.where(m, dhh, 0); dhh_prev = np.where(m, 0, dhh)
        dhc_new = np.where(m, dhc, 0); dhc_prev = np.where(m, 0, dhc)
        dr0 = dh_new * sc
        dd_h = dh_new * r0 * dsc
        do = dr0 * tc
        dc_new = dc_new + dr0 * o * (1 - tc * tc)
        dgates = np.concatenate([dc_new * g * i * (1 - i), dc_new * c_prev * f * (1 - f),
                                 dc_new * i * (1 - g * g), do * o * (1 - o)], axis=1)
        dgates = _clip(dgates, clipv)
        dc_prev = dc_prev + dc_new * f
        dxproj[t] = dgates
        g_['W_h'] += h_prev.T @ dgates
        dh_prev = dh_prev + dgates @ p['W_h'].T
        g_['W_hz'] += z_h.T @ dd_h
        dz_h = dd_h @ p['W_hz'].T
        g_['W_hhat_h'] += hh_new.T @ dz_h; g_['b_hhat_h'] += dz_h.sum(0)
        dhh_new = dhh_new + dz_h @ p['W_hhat_h'].T
        dho = dhh_new * thc
        dhc_new = dhc_new + dhh_new * ho * (1 - thc * thc)
        dhg = np.concatenate([dhc_new * hgc * hi * (1 - hi), dhc_new * hc_prev * hf * (1 - hf),
                              dhc_new * hi * (1 - hgc * hgc), dho * ho * (1 - ho)], axis=1)
        dhg = _clip(dhg, clipv)
        dhc_prev = dhc_prev + dhc_new * hf
        g_['W_xhat'] += hx.T @ dhg
        g_['bhat'] += dhg.sum(0)
        g_['W_hhat'] += hh_prev.T @ dhg
        dhx = dhg @ p['W_xhat'].T
        dh_prev = dh_prev + dhx[:, :N]
        dx[t] = dhx[:, N:]
        dhh_prev = dhh_prev + dhg @ p['W_hhat'].T
        dh, dc, dhh, dhc = dh_prev, dc_prev, dhh_prev, dhc_prev
    g_['W_x'] += x.reshape(T * B, D).T @ dxproj.reshape(T * B, 4 * N)
    g_['b'] += dxproj.sum((0, 1))
    dx += dxproj @ p['W_x'].T
    return dx, g_


# --------------------------------------------------------------------------------------------
# ScalingHyperLSTMLayer scan (libs/lasagne_libs/hyper_layers.py:850-1205) with use_peepholes=False and
# use_layer_norm=False, the configuration models/gating_hyper_nets.py:27-38 defaults to: the cell-to
# -gate vectors and the scale_outer_cell map are then non-trainable zeros (:585-588, :716-723) and
# drop out.  An "inner" LSTM (sees x_t, its own h and the outer h) emits three affine maps of its
# new hidden state that scale the outer LSTM's input term, recurrent term and bias (:1030-1040).
# grad_clip is applied per gate pre-activation AND on outcell, for both cells (:992-1022,:1057-1089).
# p keys: Wi_x (D,4H), Wi_h (H,4H), Wi_o (N,4H), bi (4H); Wo_x (D,4N), Wo_h (N,4N);
#         S_in, S_hid, S_bias (H,4N) with biases s_in, s_hid, s_bias (4N)
# --------------------------------------------------------------------------------------------
def _scal_cell_fwd(g, c_prev, peep, lnp, U, use_ln):
    """One cell of ScalingHyperLSTMLayer (hyper_layers.py:977-1025 inner, :1042-1091 outer) from the gate
    pre-activations g (B,4U), gate-major.  peep: (B or 1, 4U) what multiplies the cell state into the gates
    (vectors for the inner cell, the per-sample scale_outer_cell map for the outer one; the cell gate's block is zero).
    lnp: (alpha (5,U), beta (5,U)) for ingate, forgetgate, cell, outgate, outcell.  -> c_new, h_new, cache."""
    sl = lambda v, n: v[:, n * U:(n + 1) * U]
    pre = [sl(g, k) + c_prev * sl(peep, k) for k in range(3)]
    lc = [None] * 5
    if use_ln:
        for k in range(3):
            pre[k], lc[k] = ln_fwd(pre[k], lnp[0][k], lnp[1][k])
    i = _sig(pre[0]); f = _sig(pre[1]); gc = np.tanh(pre[2])
    c_new = f * c_prev + i * gc
    po = sl(g, 3) + c_new * sl(peep, 3)
    oc = c_new
    if use_ln:
        po, lc[3] = ln_fwd(po, lnp[0][3], lnp[1][3])
        oc, lc[4] = ln_fwd(c_new, lnp[0][4], lnp[1][4])
    o = _sig(po); tc = np.tanh(oc)
    return c_new, o * tc, (i, f, gc, o, tc, c_new, lc)


def _scal_cell_bwd(dh, dc_in, c_prev, peep, lnp, cache, U, use_ln, clipv):
    """-> dg (B,4U) wrt the gate pre-activations, dc_prev, dpeep (B,4U) wrt the peephole factors (per sample),
    dln = (dalpha (5,B,U), dbeta (5,B,U)) per sample (zeros without layer norm)."""
    i, f, gc, o, tc, c_new, lc = cache
    sl = lambda v, n: v[:, n * U:(n + 1) * U]
    B = dh.shape[0]
    dal = np.zeros((5, B, U), dh.dtype); dbe = np.zeros((5, B, U), dh.dtype)

    def back_ln(d, k):                                     # d: gradient at the (clipped) LN output
        d = _clip(d, clipv)
        if not use_ln:
            return d
        dx, da, db = ln_bwd(d, lnp[0][k], lc[k])
        dal[k] = da; dbe[k] = db
        return dx
    dpo = back_ln(dh * tc * o * (1 - o), 3)
    dc = dc_in + back_ln(dh * o * (1 - tc * tc), 4) + dpo * sl(peep, 3)
    dpi = back_ln(dc * gc * i * (1 - i), 0)
    dpf = back_ln(dc * c_prev * f * (1 - f), 1)
    dpc = back_ln(dc * i * (1 - gc * gc), 2)
    dc_prev = dc * f + dpi * sl(peep, 0) + dpf * sl(peep, 1) + dpc * sl(peep, 2)
    dg = np.concatenate([dpi, dpf, dpc, dpo], axis=1)
    dpeep = np.concatenate([dpi * c_prev, dpf * c_prev, dpc * c_prev, dpo * c_new], axis=1)
    return dg, dc_prev, dpeep, (dal, dbe)


def scaling_scan_fwd(x, mask, p, spec):
    T, B, D = x.shape
    N, H = spec['N'], spec['H']
    U = spec.get('U', H)                                  # inner cell width: H, or N for EqualScaleHyperLSTMLayer (:1258-1292)
    use_ln, peep = spec.get('ln', False), spec.get('peep', False)
    dt = x.dtype
    inner_in = x @ p['Wi_x'] + p['bi']                    # :951
    outer_in = x @ p['Wo_x']                              # :952
    one = np.ones((B, 1), dt)
    ic = one @ p['ic0']; ih = one @ p['ih0']; oc = one @ p['oc0']; oh = one @ p['oh0']   # :1127-1131
    order = range(T - 1, -1, -1) if spec.get('backwards') else range(T)
    hid_i = np.zeros((T, B, H), dt); hid_o = np.zeros((T, B, N), dt)
    cache = [None] * T
    zeroN = np.zeros((B, 4 * N), dt)
    for t in order:
        m = mask[t][:, None] != 0
        ig = inner_in[t] + ih @ p['Wi_h'] + oh @ p['Wi_o']            # :973-975
        ic_new, u_new, cin = _scal_cell_fwd(ig, ic, p['wpeep'][None, :], (p['ln_ia'], p['ln_ib']), U, use_ln)
        ih_new = u_new @ p['P'] if 'P' in p else u_new                # EqualScale: inner_hid . W_inner_hid_prj (:1709)
        s_in = ih_new @ p['S_in'] + p['s_in']                         # :1030
        s_hid = ih_new @ p['S_hid'] + p['s_hid']                      # :1031
        s_cl = ih_new @ p['S_cell'] + p['s_cell'] if peep else zeroN  # :1032
        s_b = ih_new @ p['S_bias'] + p['s_bias']                      # :1033
        a = oh @ p['Wo_h']
        og = s_in * outer_in[t] + s_hid * a + s_b                     # :1038-1040
        oc_new, oh_new, cout = _scal_cell_fwd(og, oc, s_cl, (p['ln_oa'], p['ln_ob']), N, use_ln)
        cache[t] = (m, ic, ih, oc, oh, ih_new, s_in, s_hid, s_cl, a, cin, cout, u_new)
        ic = np.where(m, ic_new, ic); ih = np.where(m, ih_new, ih)    # step_masked :1097-1118
        oc = np.where(m, oc_new, oc); oh = np.where(m, oh_new, oh)
        hid_i[t] = ih; hid_o[t] = oh
    return hid_i, hid_o, (cache, inner_in, outer_in)


def scaling_scan_bwd(dhid_i, dhid_o, x, fwd_cache, p, spec):
    cache, inner_in, outer_in = fwd_cache
    T, B, D = x.shape
    N, H = spec['N'], spec['H']
    U = spec.get('U', H)
    use_ln, peep = spec.get('ln', False), spec.get('peep', False)
    clipv = spec.get('grad_clip', 0)
    dt = dhid_o.dtype
    order = list(range(T - 1, -1, -1) if spec.get('backwards') else range(T))
    g_ = {k: np.zeros_like(v) for k, v in p.items()}
    doh = np.zeros((B, N), dt); doc = np.zeros((B, N), dt)
    dih = np.zeros((B, H), dt); dic = np.zeros((B, U), dt)
    d_inner_in = np.zeros((T, B, 4 * U), dt); d_outer_in = np.zeros((T, B, 4 * N), dt)
    for t in reversed(order):
        (m, ic_prev, ih_prev, oc_prev, oh_prev, ih_new, s_in, s_hid, s_cl, a, cin, cout, u_new) = cache[t]
        doh_t = dhid_o[t] + doh
        dih_t = dhid_i[t] + dih
        doh_new = np.where(m, doh_t, 0); doh_prev = np.where(m, 0, doh_t)
        doc_new = np.where(m, doc, 0); doc_prev = np.where(m, 0, doc)
        dih_new = np.where(m, dih_t, 0); dih_prev = np.where(m, 0, dih_t)
        dic_new = np.where(m, dic, 0); dic_prev = np.where(m, 0, dic)
        # outer cell (:1042-1091)
        dog, dcp, ds_cl, (dal, dbe) = _scal_cell_bwd(doh_new, doc_new, oc_prev, s_cl, (p['ln_oa'], p['ln_ob']), cout, N,
                                                     use_ln, clipv)
        doc_prev = doc_prev + dcp
        g_['ln_oa'] += dal.sum(1); g_['ln_ob'] += dbe.sum(1)
        ds_in = dog * outer_in[t]; d_outer_in[t] = dog * s_in
        ds_hid = dog * a; da = dog * s_hid
        ds_b = dog
        g_['Wo_h'] += oh_prev.T @ da
        doh_prev = doh_prev + da @ p['Wo_h'].T
        maps = [('in', ds_in), ('hid', ds_hid), ('bias', ds_b)] + ([('cell', ds_cl)] if peep else [])
        for nm, dsv in maps:
            g_['S_' + nm] += ih_new.T @ dsv
            g_['s_' + nm] += dsv.sum(0)
            dih_new = dih_new + dsv @ p['S_' + nm].T
        # inner cell (:977-1025)
        du = dih_new
        if 'P' in p:
            g_['P'] += u_new.T @ dih_new
            du = dih_new @ p['P'].T
        dig, dcp, dwp, (dal, dbe) = _scal_cell_bwd(du, dic_new, ic_prev, p['wpeep'][None, :], (p['ln_ia'], p['ln_ib']),
                                                   cin, U, use_ln, clipv)
        dic_prev = dic_prev + dcp
        g_['wpeep'] += dwp.sum(0)
        g_['ln_ia'] += dal.sum(1); g_['ln_ib'] += dbe.sum(1)
        d_inner_in[t] = dig
        g_['Wi_h'] += ih_prev.T @ dig
        g_['Wi_o'] += oh_prev.T @ dig
        dih_prev = dih_prev + dig @ p['Wi_h'].T
        doh_prev = doh_prev + dig @ p['Wi_o'].T
        doh, doc, dih, dic = doh_prev, doc_prev, dih_prev, dic_prev
    g_['ic0'] += dic.sum(0, keepdims=True); g_['ih0'] += dih.sum(0, keepdims=True)
    g_['oc0'] += doc.sum(0, keepdims=True); g_['oh0'] += doh.sum(0, keepdims=True)
    xf = x.reshape(T * B, D)
    g_['Wi_x'] += xf.T @ d_inner_in.reshape(T * B, -1)
    g_['bi'] += d_inner_in.sum((0, 1))
    g_['Wo_x'] += xf.T @ d_outer_in.reshape(T * B, -1)
    dx = d_inner_in @ p['Wi_x'].T + d_outer_in @ p['Wo_x'].T
    return dx, g_


# --------------------------------------------------------------------------------------------
# ProjectionHyperLSTMLayer scan (libs/lasagne_libs/hyper_layers.py:14-523, step :319-417): two N-unit
# LSTMP cells with cell layer norm, both projected to F = num_factors.  The inner cell sees x_t, its own
# projected hid and the outer projected hid (:329-331); three affine maps of the NEW inner hid scale the
# outer cell's input term, recurrent term and bias (:367-377).  grad_clip on each gate pre-activation
# and on each outcell (:338-350,358-361,384-396,404-407).  Output = [inner_hid | outer_hid] (:519).
# p keys: Wi_x (D,4N), Wi_i (F,4N), Wi_o (F,4N), bi (4N), Wi_p (N,F), ai, bi_c (N,)  [inner cell LN]
#         Wo_x (D,4N), Wo_o (F,4N), S (F,12N) = [to_in | to_outer | to_bias], s (12N), Wo_p (N,F), ao, bo_c
# --------------------------------------------------------------------------------------------
def prjhyper_scan_fwd(x, mask, p, spec):
    T, B, D = x.shape
    N, F = spec['N'], spec['F']
    dt = x.dtype
    inner_in = x @ p['Wi_x'] + p['bi']                    # :316
    outer_in = x @ p['Wo_x']                              # :317
    ic = np.zeros((B, N), dt); ih = np.zeros((B, F), dt)
    oc = np.zeros((B, N), dt); oh = np.zeros((B, F), dt)
    order = range(T - 1, -1, -1) if spec.get('backwards') else range(T)
    hid_i = np.zeros((T, B, F), dt); hid_o = np.zeros((T, B, F), dt)
    cache = [None] * T
    for t in order:
        m = mask[t][:, None] != 0
        ig = inner_in[t] + ih @ p['Wi_i'] + oh @ p['Wi_o']                        # :329-331
        ii = _sig(ig[:, 0:N]); if_ = _sig(ig[:, N:2 * N]); igc = np.tanh(ig[:, 2 * N:3 * N]); io = _sig(ig[:, 3 * N:4 * N])
        ic_new = if_ * ic + ii * igc                                              # :355
        iout, iln = ln_fwd(ic_new, p['ai'], p['bi_c'])                            # :357
        itc = np.tanh(iout)
        ir = io * itc                                                             # :363
        ih_new = ir @ p['Wi_p']                                                   # :364
        sc = ih_new @ p['S'] + p['s']                                             # :367
        s_in, s_hid, s_b = sc[:, 0:4 * N], sc[:, 4 * N:8 * N], sc[:, 8 * N:12 * N]
        a = oh @ p['Wo_o']
        og = outer_in[t] * s_in + a * s_hid + s_b                                 # :375-377
        oi = _sig(og[:, 0:N]); of = _sig(og[:, N:2 * N]); ogc = np.tanh(og[:, 2 * N:3 * N]); oo = _sig(og[:, 3 * N:4 * N])
        oc_new = of * oc + oi * ogc                                               # :401
        oout, oln = ln_fwd(oc_new, p['ao'], p['bo_c'])                            # :403
        otc = np.tanh(oout)
        orr = oo * otc                                                            # :409
        oh_new = orr @ p['Wo_p']                                                  # :410
        cache[t] = (m, ic, ih, oc, oh, ii, if_, igc, io, iln, itc, ir, ih_new, s_in, s_hid, a,
                    oi, of, ogc, oo, oln, otc, orr)
        ic = np.where(m, ic_new, ic); ih = np.where(m, ih_new, ih)                # :441-444
        oc = np.where(m, oc_new, oc); oh = np.where(m, oh_new, oh)
        hid_i[t] = ih; hid_o[t] = oh
    return hid_i, hid_o, (cache, inner_in, outer_in)


def prjhyper_scan_bwd(dhid_i, dhid_o, x, fwd_cache, p, spec):
    cache, inner_in, outer_in = fwd_cache
    T, B, D = x.shape
    N, F = spec['N'], spec['F']
    clipv = spec.get('grad_clip', 0)
    dt = dhid_o.dtype
    order = list(range(T - 1, -1, -1) if spec.get('backwards') else range(T))
    g_ = {k: np.zeros_like(v) for k, v in p.items()}
    doh = np.zeros((B, F), dt); doc = np.zeros((B, N), dt)
    dih = np.zeros((B, F), dt); dic = np.zeros((B, N), dt)
    d_ii = np.zeros((T, B, 4 * N), dt); d_oi = np.zeros((T, B, 4 * N), dt)
    for t in reversed(order):
        (m, ic_prev, ih_prev, oc_prev, oh_prev, ii, if_, igc, io, iln, itc, ir, ih_new, s_in, s_hid, a,
         oi, of, ogc, oo, oln, otc, orr) = cache[t]
        doh_t = dhid_o[t] + doh; dih_t = dhid_i[t] + dih
        doh_new = np.where(m, doh_t, 0); doh_prev = np.where(m, 0, doh_t)
        doc_new = np.where(m, doc, 0); doc_prev = np.where(m, 0, doc)
        dih_new = np.where(m, dih_t, 0); dih_prev = np.where(m, 0, dih_t)
        dic_new = np.where(m, dic, 0); dic_prev = np.where(m, 0, dic)
        # outer cell
        g_['Wo_p'] += orr.T @ doh_new
        dorr = doh_new @ p['Wo_p'].T
        doo = dorr * otc
        dout = _clip(dorr * oo * (1 - otc * otc), clipv)
        dcl, da_, db_ = ln_bwd(dout, p['ao'], oln)
        g_['ao'] += da_.sum(0); g_['bo_c'] += db_.sum(0)
        doc_new = doc_new + dcl
        dog = np.concatenate([doc_new * ogc * oi * (1 - oi), doc_new * oc_prev * of * (1 - of),
                              doc_new * oi * (1 - ogc * ogc), doo * oo * (1 - oo)], axis=1)
        dog = _clip(dog, clipv)
        doc_prev = doc_prev + doc_new * of
        d_oi[t] = dog * s_in
        dsc = np.concatenate([dog * outer_in[t], dog * a, dog], axis=1)
        da = dog * s_hid
        g_['Wo_o'] += oh_prev.T @ da
        doh_prev = doh_prev + da @ p['Wo_o'].T
        g_['S'] += ih_new.T @ dsc; g_['s'] += dsc.sum(0)
        dih_new = dih_new + dsc @ p['S'].T
        # inner cell
        g_['Wi_p'] += ir.T @ dih_new
        dir_ = dih_new @ p['Wi_p'].T
        dio = dir_ * itc
        diout = _clip(dir_ * io * (1 - itc * itc), clipv)
        dcl, da_, db_ = ln_bwd(diout, p['ai'], iln)
        g_['ai'] += da_.sum(0); g_['bi_c'] += db_.sum(0)
        dic_new = dic_new + dcl
        dig = np.concatenate([dic_new * igc * ii * (1 - ii), dic_new * ic_prev * if_ * (1 - if_),
                              dic_new * ii * (1 - igc * igc), dio * io * (1 - io)], axis=1)
        dig = _clip(dig, clipv)
        dic_prev = dic_prev + dic_new * if_
        d_ii[t] = dig
        g_['Wi_i'] += ih_prev.T @ dig; g_['Wi_o'] += oh_prev.T @ dig
        dih_prev = dih_prev + dig @ p['Wi_i'].T
        doh_prev = doh_prev + dig @ p['Wi_o'].T
        doh, doc, dih, dic = doh_prev, doc_prev, dih_prev, dic_prev
    xf = x.reshape(T * B, D)
    g_['Wi_x'] += xf.T @ d_ii.reshape(T * B, -1); g_['bi'] += d_ii.sum((0, 1))
    g_['Wo_x'] += xf.T @ d_oi.reshape(T * B, -1)
    dx = d_ii @ p['Wi_x'].T + d_oi @ p['Wo_x'].T
    return dx, g_


# --------------------------------------------------------------------------------------------
# Layer wrappers.  Each takes x (B,T,D) batch-major like the reference layers, a parameter dict
# keyed by the reference's add_param names, and returns (B,T,out) plus a cache for backward.
# --------------------------------------------------------------------------------------------
GATES_LHUC = ('ig', 'fg', 'c', 'og')                       # libs/lhuclstm_layers.py:53-56
GATES_REC = ('ingate', 'forgetgate', 'cell', 'outgate')    # recurrent_layers.py:341-359


def _tm(x):       # dimshuffle(1,0,2)
    return np.ascontiguousarray(np.swapaxes(x, 0, 1))


def _stack(P, fmt, gates, axis=1):
    return np.concatenate([P[fmt.format(g)] for g in gates], axis=axis)


def _unstack_into(G, fmt, gates, val, axis=1):
    n = val.shape[axis] // 4
    for k, g in enumerate(gates):
        idx = [slice(None)] * val.ndim
        idx[axis] = slice(k * n, (k + 1) * n)
        G[fmt.format(g)] = val[tuple(idx)]


class OracleLayer(object):
    """Base: forward(x, mask, cond) -> y ; backward(dy) -> dx, fills self.grads (name -> array)."""
    def forward(self, x, mask, cond=None):
        raise NotImplementedError

    def backward(self, dy):
        raise NotImplementedError


class LSTMLayerO(OracleLayer):
    """libs/lhuclstm_layers.py:141-218 (LSTMLayer) and :220-324 (TBPTTLSTMLayer when stateful)."""
    def __init__(self, P, num_units, num_proj_units=0, use_layer_norm=False, grad_clipping=0,
                 backwards=False, stateful=False, context=0):
        self.P = P; self.N = num_units; self.Pj = num_proj_units; self.ln = use_layer_norm
        self.spec = dict(N=num_units, P=num_proj_units, ln='static' if use_layer_norm else 'none',
                         cell_ln=use_layer_norm, grad_clip=grad_clipping, backwards=backwards)
        self.stateful = stateful; self.context = context
        self.c0 = None; self.h0 = None

    def reset(self, batch_size):                        # :221-228
        self.c0 = np.zeros((batch_size, self.N)); self.h0 = np.zeros((batch_size, self.Pj or self.N))

    def _p(self):
        P = self.P
        p = dict(W_h=_stack(P, 'W_h_{}', GATES_LHUC), b=_stack(P, 'b_{}', GATES_LHUC, axis=0),
                 W_p=P.get('W_p'))
        if self.ln:                                     # :77-78 alpha only, :101 cell LN alpha+beta
            p.update(alpha_in=P['W_x_alpha'], alpha_hid=P['W_h_alpha'], alpha_c=P['W_c_alpha'],
                     beta_c=P['W_c_beta'])
        return p

    def forward(self, x, mask, cond=None):
        self.x = _tm(x); m = _tm(mask[:, :, None])[:, :, 0]
        self.Wx = _stack(self.P, 'W_x_{}', GATES_LHUC)
        xin = self.x @ self.Wx                          # :189
        self.p = self._p()
        c0 = self.c0 if self.stateful else None
        h0 = self.h0 if self.stateful else None
        hid, cell, self.cache = lstm_scan_fwd(xin, m, self.p, self.spec, c0, h0)
        if self.stateful:                               # :313-319
            T = hid.shape[0]
            last = -1
            if self.context:
                last = (T if T < self.context else self.context) - 1
            self.c0 = cell[last].copy(); self.h0 = hid[last].copy()
        return _tm(hid)

    def backward(self, dy):
        dxin, g, _ = lstm_scan_bwd(_tm(dy), self.cache, self.p, self.spec)
        T, B, D = self.x.shape
        G = {}
        _unstack_into(G, 'W_h_{}', GATES_LHUC, g['W_h'])
        _unstack_into(G, 'b_{}', GATES_LHUC, g['b'], axis=0)
        _unstack_into(G, 'W_x_{}', GATES_LHUC, self.x.reshape(T * B, D).T @ dxin.reshape(T * B, -1))
        if self.Pj:
            G['W_p'] = g['W_p']
        if self.ln:
            G.update(W_x_alpha=g['alpha_in'], W_h_alpha=g['alpha_hid'], W_c_alpha=g['alpha_c'],
                     W_c_beta=g['beta_c'])
        self.grads = G
        return _tm(dxin @ self.Wx.T)


class LHUCLSTMLayerO(LSTMLayerO):
    """libs/lhuclstm_layers.py:326-471 SpeakerLHUCLSTMLayer with the i-vector embedding (:491-510):
    A1 recurrence, then hid_out *= relu(MLP(embedding)) after the scan (:461-462)."""
    def __init__(self, P, num_units, num_pred_units, num_pred_layers, **kw):
        LSTMLayerO.__init__(self, P, num_units, **kw)
        self.npl = num_pred_layers

    def forward(self, x, mask, cond=None):
        hid = LSTMLayerO.forward(self, x, mask)
        emb = cond[:, 0, :]                             # :504-510 first frame of the i-vector
        acts = [emb]
        for i in range(self.npl):                       # :405-407
            acts.append(np.maximum(acts[-1] @ self.P['W_e_h_%d' % i] + self.P['b_e_h_%d' % i], 0))
        sf = acts[-1] @ self.P['W_e_h_%d' % self.npl] + self.P['b_e_h_%d' % self.npl]   # :409
        self.acts = acts; self.sf = sf; self.hid = hid
        return hid * np.maximum(sf, 0)[:, None, :]      # :462

    def backward(self, dy):
        dsf = (dy * self.hid).sum(1) * (self.sf > 0)
        dhid = dy * np.maximum(self.sf, 0)[:, None, :]
        dx = LSTMLayerO.backward(self, dhid)
        G = self.grads
        G['W_e_h_%d' % self.npl] = self.acts[-1].T @ dsf
        G['b_e_h_%d' % self.npl] = dsf.sum(0)
        da = dsf @ self.P['W_e_h_%d' % self.npl].T
        for i in range(self.npl - 1, -1, -1):
            da = da * (self.acts[i + 1] > 0)
            G['W_e_h_%d' % i] = self.acts[i].T @ da
            G['b_e_h_%d' % i] = da.sum(0)
            da = da @ self.P['W_e_h_%d' % i].T
        self.demb = da                                  # gradient wrt the speaker embedding
        return dx


class ProjectLSTMLayerO(OracleLayer):
    """libs/lasagne_libs/recurrent_layers.py:290-546 (no peepholes, learn_init False)."""
    ln = 'none'

    def __init__(self, P, num_units, num_factors=None, grad_clipping=0, backwards=False):
        self.P = P; self.N = num_units; self.F = num_factors or num_units // 2
        self.spec = dict(N=num_units, P=self.F, ln=self.ln, cell_ln=(self.ln == 'static'),
                         grad_clip=grad_clipping, backwards=backwards)

    def _p(self):
        return dict(W_h=_stack(self.P, 'W_hid_to_{}', GATES_REC), W_p=self.P['W_hid_prj'],
                    b=_stack(self.P, 'b_{}', GATES_REC, axis=0))

    def forward(self, x, mask, cond=None):
        self.x = _tm(x); m = _tm(mask[:, :, None])[:, :, 0]
        self.Wx = _stack(self.P, 'W_in_to_{}', GATES_REC)
        xin = self.x @ self.Wx                          # :431 (+b_stacked: same sum, added in step)
        self.p = self._p()
        c0, h0 = getattr(self, 'init', None) or (None, None)     # tests only: start a window from a given state
        hid, self.cell_seq, self.cache = lstm_scan_fwd(xin, m, self.p, self.spec, c0, h0)
        self.hid_seq = hid
        return _tm(hid)

    def backward(self, dy):
        dxin, g, _ = lstm_scan_bwd(_tm(dy), self.cache, self.p, self.spec)
        T, B, D = self.x.shape
        G = {'W_hid_prj': g['W_p']}
        _unstack_into(G, 'W_hid_to_{}', GATES_REC, g['W_h'])
        _unstack_into(G, 'b_{}', GATES_REC, g['b'], axis=0)
        _unstack_into(G, 'W_in_to_{}', GATES_REC, self.x.reshape(T * B, D).T @ dxin.reshape(T * B, -1))
        self._extra_grads(G, g)
        self.grads = G
        return _tm(dxin @ self.Wx.T)

    def _extra_grads(self, G, g):
        pass


class LayerNormProjectLSTMLayerO(ProjectLSTMLayerO):
    """libs/lasagne_libs/recurrent_layers.py:548-814: ln(x_t;a,b)+ln(hW;a,b)+b_stacked, cell LN."""
    ln = 'static'

    def _p(self):
        p = ProjectLSTMLayerO._p(self)
        P = self.P
        p.update(alpha_in=P['W_alpha_in'], beta_in=P['W_beta_in'], alpha_hid=P['W_alpha_hid'],
                 beta_hid=P['W_beta_hid'], alpha_c=P['W_alpha_cell'], beta_c=P['W_beta_cell'])
        return p

    def _extra_grads(self, G, g):
        G.update(W_alpha_in=g['alpha_in'], W_beta_in=g['beta_in'], W_alpha_hid=g['alpha_hid'],
                 W_beta_hid=g['beta_hid'], W_alpha_cell=g['alpha_c'], W_beta_cell=g['beta_c'])


class InputCondLayerNormProjectLSTMLayerO(OracleLayer):
    """libs/lasagne_libs/recurrent_layers.py:1075-1338: per-sample alpha/beta from the i-vector."""
    def __init__(self, P, num_units, num_factors, grad_clipping=0, backwards=False):
        self.P = P; self.N = num_units; self.F = num_factors
        self.spec = dict(N=num_units, P=num_units // 2, ln='sample', cell_ln=False,
                         grad_clip=grad_clipping, backwards=backwards)

    def forward(self, x, mask, cond=None):
        P = self.P
        pre = np.tanh(cond @ P['W_cond_prj'] + P['b_cond_prj'])         # :1209-1210
        msum = mask.sum(axis=1, keepdims=True)
        s = (pre * mask[:, :, None]).sum(axis=1) / msum                 # :1211
        self.cond = cond; self.pre = pre; self.maskbt = mask; self.msum = msum; self.s = s
        alpha_in = s @ P['W_alpha_in'] + P['b_alpha_in']                # :1215
        alpha_hid = s @ P['W_alpha_hid'] + P['b_alpha_hid']             # :1216
        beta = s @ P['W_beta'] + P['b_beta']                            # :1217
        self.x = _tm(x); m = _tm(mask[:, :, None])[:, :, 0]
        self.Wx = _stack(P, 'W_in_to_{}', GATES_REC)
        xin = self.x @ self.Wx + beta[None]                             # :1233
        self.p = dict(W_h=_stack(P, 'W_hid_to_{}', GATES_REC), W_p=P['W_hid_prj'], b=None,
                      alpha_in=alpha_in, alpha_hid=alpha_hid)
        c0, h0 = getattr(self, 'init', None) or (None, None)     # tests only: start a window from a given state
        hid, self.cell_seq, self.cache = lstm_scan_fwd(xin, m, self.p, self.spec, c0, h0)
        self.hid_seq = hid
        return _tm(hid)

    def get_sample_feat(self):                                          # :1337-1338
        return self.s

    def backward(self, dy):
        P = self.P
        dxin, g, _ = lstm_scan_bwd(_tm(dy), self.cache, self.p, self.spec)
        T, B, D = self.x.shape
        G = {'W_hid_prj': g['W_p']}
        _unstack_into(G, 'W_hid_to_{}', GATES_REC, g['W_h'])
        _unstack_into(G, 'W_in_to_{}', GATES_REC, self.x.reshape(T * B, D).T @ dxin.reshape(T * B, -1))
        dbeta = dxin.sum(0)                                             # (B,4N)
        dal_in, dal_hid = g['alpha_in'], g['alpha_hid']
        s = self.s
        G['W_alpha_in'] = s.T @ dal_in; G['b_alpha_in'] = dal_in.sum(0)
        G['W_alpha_hid'] = s.T @ dal_hid; G['b_alpha_hid'] = dal_hid.sum(0)
        G['W_beta'] = s.T @ dbeta; G['b_beta'] = dbeta.sum(0)
        ds = dal_in @ P['W_alpha_in'].T + dal_hid @ P['W_alpha_hid'].T + dbeta @ P['W_beta'].T
        dpre = (ds / self.msum)[:, None, :] * self.maskbt[:, :, None] * (1 - self.pre ** 2)
        Bc, Tc, I = self.cond.shape
        G['W_cond_prj'] = self.cond.reshape(Bc * Tc, I).T @ dpre.reshape(Bc * Tc, -1)
        G['b_cond_prj'] = dpre.sum((0, 1))
        self.grads = G
        return _tm(dxin @ self.Wx.T)


class RecLayerO(OracleLayer):
    """Every LSTM layer class of libs/lasagne_libs/recurrent_layers.py as one parameterised
    restatement (variant table: SURVEY.md Appendix A; the table itself is passed in by the caller as
    `variant`, keys documented in asr_amd/specs.py REC_CLASSES, so that the oracle imports nothing
    from the product).  Line references per variant key:
      cond prologue, masked mean of tanh  :944-946 (own input), :1209-1211 (external)
      cond prologue, unmasked max of relu :1468-1470, :1733-1735
      alpha_in / alpha_hid / beta (B,4N)  :950-952, :1215-1217;  beta = s.W_beta + b_stacked :2807
      beta before LN (added to x.W)       :968, :1233;   beta after LN :2824, :4937
      static LN + b_stacked               :711-713, :2276-2278;  per-gate LN :2531-2532, :3113-3114
      cell LN                             :740, :2305, :2849;    projection :748"""
    def __init__(self, P, variant, num_units, out_width=0, grad_clipping=0, backwards=False, cls=None,
                 peepholes=False, learn_init=False, only_return_final=False, only_return_hidden=True):
        """Options of the classes (:52-57): peepholes (LSTMLayer, ProjectLSTMLayer, *_v0_1 plain classes), learn_init
        (every class without conditioning: cell_init (1,N), hid_init (1,K) rows broadcast over the batch, :246-248),
        only_return_final (hid_out[-1] in scan order, :265-266), only_return_hidden=False (cell_out appended,
        :275-288)."""
        self.P = P; self.v = variant; self.N = num_units; self.Pw = out_width
        self.peepholes = peepholes; self.learn_init = learn_init
        self.only_return_final = only_return_final; self.only_return_hidden = only_return_hidden
        v = variant
        self.spec = dict(N=num_units, P=out_width, ln=v['ln'], per_gate=v['per_gate'], cell_ln=v['cell_ln'],
                         grad_clip=grad_clipping, backwards=backwards)

    def forward(self, x, mask, cond=None):
        P = self.P; v = self.v
        self.x = _tm(x); m = _tm(mask[:, :, None])[:, :, 0]
        self.Wx = _stack(P, 'W_in_to_{}', GATES_REC)
        xin = self.x @ self.Wx
        p = dict(W_h=_stack(P, 'W_hid_to_{}', GATES_REC), W_p=P.get('W_hid_prj'), b=None)
        b_st = _stack(P, 'b_{}', GATES_REC, axis=0) if v['bias'] else None
        if v['cond']:
            src = x if v['cond'] == 'own' else cond
            a = src @ P['W_cond_prj'] + P['b_cond_prj']
            if v['pool'] == 'mean_tanh':
                pre = np.tanh(a)
                msum = mask.sum(axis=1, keepdims=True)
                sfeat = (pre * mask[:, :, None]).sum(axis=1) / msum
                self.msum = msum
            else:
                pre = np.maximum(a, 0)
                sfeat = pre.max(axis=1)
            self.src = src; self.a = a; self.pre = pre; self.maskbt = mask; self.s = sfeat
            p['alpha_in'] = sfeat @ P['W_alpha_in'] + P['b_alpha_in']
            p['alpha_hid'] = sfeat @ P['W_alpha_hid'] + P['b_alpha_hid']
            beta = sfeat @ P['W_beta'] + (b_st if v['bias'] else P['b_beta'])
            if v['beta'] == 'pre':
                xin = xin + beta[None]
            else:
                p['b'] = beta
        else:
            p['b'] = b_st
            if v['ln'] == 'static':
                p['alpha_in'] = P['W_alpha_in']; p['alpha_hid'] = P['W_alpha_hid']
                if v['ln_beta']:
                    p['beta_in'] = P['W_beta_in']; p['beta_hid'] = P['W_beta_hid']
        if v['cell_ln']:
            p['alpha_c'] = P['W_alpha_cell']
            p['beta_c'] = P['num_factors'] if v['cond'] else P['W_beta_cell']
        if self.peepholes:
            p['peep'] = (P['W_cell_to_ingate'], P['W_cell_to_forgetgate'], P['W_cell_to_outgate'])
        self.p = p
        c0 = h0 = None
        if self.learn_init:
            ones = np.ones((x.shape[0], 1), xin.dtype)
            c0 = ones @ P['cell_init']; h0 = ones @ P['hid_init']
        hid, cell, self.cache = lstm_scan_fwd(xin, m, p, self.spec, c0, h0)
        last = 0 if self.spec['backwards'] else hid.shape[0] - 1
        self._shape = hid.shape
        if self.only_return_final:
            hid, cell = hid[last], cell[last]
        else:
            hid, cell = _tm(hid), _tm(cell)
        return hid if self.only_return_hidden else np.concatenate([hid, cell], axis=-1)

    def get_sample_feat(self):
        return self.s

    def backward(self, dy):
        P = self.P; v = self.v
        Tn, Bn, Pn = self._shape
        dcell = None
        if not self.only_return_hidden:
            dy, dcell = dy[..., :Pn], dy[..., Pn:]
        if self.only_return_final:
            last = 0 if self.spec['backwards'] else Tn - 1
            full = np.zeros((Tn, Bn, Pn), dy.dtype); full[last] = dy; dy = full
            if dcell is not None:
                full = np.zeros((Tn, Bn, self.N), dy.dtype); full[last] = dcell; dcell = full
        else:
            dy = _tm(dy)
            dcell = _tm(dcell) if dcell is not None else None
        dxin, g, (dc0, dh0) = lstm_scan_bwd(dy, self.cache, self.p, self.spec, dcell)
        T, B, D = self.x.shape
        G = {}
        if self.peepholes:
            G['W_cell_to_ingate'], G['W_cell_to_forgetgate'], G['W_cell_to_outgate'] = g['peep']
        if self.learn_init:
            G['cell_init'] = dc0.sum(0, keepdims=True); G['hid_init'] = dh0.sum(0, keepdims=True)
        _unstack_into(G, 'W_hid_to_{}', GATES_REC, g['W_h'])
        _unstack_into(G, 'W_in_to_{}', GATES_REC, self.x.reshape(T * B, D).T @ dxin.reshape(T * B, -1))
        if self.Pw:
            G['W_hid_prj'] = g['W_p']
        dx = _tm(dxin @ self.Wx.T)
        if v['cell_ln']:
            G['W_alpha_cell'] = g['alpha_c']
            G['num_factors' if v['cond'] else 'W_beta_cell'] = g['beta_c']
        if v['cond']:
            dbeta = dxin.sum(0) if v['beta'] == 'pre' else g['b']
            dal_in, dal_hid = g['alpha_in'], g['alpha_hid']
            sf = self.s
            G['W_alpha_in'] = sf.T @ dal_in; G['b_alpha_in'] = dal_in.sum(0)
            G['W_alpha_hid'] = sf.T @ dal_hid; G['b_alpha_hid'] = dal_hid.sum(0)
            G['W_beta'] = sf.T @ dbeta
            if v['bias']:
                _unstack_into(G, 'b_{}', GATES_REC, dbeta.sum(0), axis=0)
            else:
                G['b_beta'] = dbeta.sum(0)
            ds = dal_in @ P['W_alpha_in'].T + dal_hid @ P['W_alpha_hid'].T + dbeta @ P['W_beta'].T
            fr = getattr(self, 'feat_reg', 0.0)     # d/ds of fr * (-mean_j var_b s_bj): the scripts' train_feat_loss
            if fr:                                  # (scripts/dln/ted/exp_prj_lstm_dln.py:120-131)
                ds = ds + fr * (-2.0 / (sf.shape[0] * sf.shape[1])) * (sf - sf.mean(0, keepdims=True))
            if v['pool'] == 'mean_tanh':
                da = (ds / self.msum)[:, None, :] * self.maskbt[:, :, None] * (1 - self.pre ** 2)
            else:      # T.max gradient goes to every position equal to the max, then relu'
                da = ds[:, None, :] * (self.pre == self.s[:, None, :]) * (self.a > 0)
            Bc, Tc, I = self.src.shape
            G['W_cond_prj'] = self.src.reshape(Bc * Tc, I).T @ da.reshape(Bc * Tc, -1)
            G['b_cond_prj'] = da.sum((0, 1))
            if v['cond'] == 'own':
                dx = dx + da @ P['W_cond_prj'].T
        else:
            if v['bias']:
                _unstack_into(G, 'b_{}', GATES_REC, g['b'], axis=0)
            if v['ln'] == 'static':
                G['W_alpha_in'] = g['alpha_in']; G['W_alpha_hid'] = g['alpha_hid']
                if v['ln_beta']:
                    G['W_beta_in'] = g['beta_in']; G['W_beta_hid'] = g['beta_hid']
        self.grads = G
        return dx


class HyperLSTMLayerO(OracleLayer):
    """libs/lasagne_libs/hyper_lhuc_layers.py:168-507."""
    def __init__(self, P, num_units, num_hyper_units, num_proj_units, grad_clipping=0, backwards=False):
        self.P = P
        self.spec = dict(N=num_units, H=num_hyper_units, Z=num_proj_units, grad_clip=grad_clipping,
                         backwards=backwards)

    def forward(self, x, mask, cond=None):
        P = self.P; G4 = GATES_LHUC
        self.p = dict(
            W_xhat=_stack(P, 'W_xhat_{}', G4), bhat=_stack(P, 'bhat_{}', G4, 0),
            W_hhat=_stack(P, 'W_hhat_{}', G4),
            W_hhat_h=_stack(P, 'W_hhat_h_{}', G4), b_hhat_h=_stack(P, 'b_hhat_h_{}', G4, 0),
            W_hhat_x=_stack(P, 'W_hhat_x_{}', G4), b_hhat_x=_stack(P, 'b_hhat_x_{}', G4, 0),
            W_hhat_b=_stack(P, 'W_hhat_b_{}', G4),
            W_hz=np.stack([P['W_hz_' + g] for g in G4]), W_xz=np.stack([P['W_xz_' + g] for g in G4]),
            W_bz=np.stack([P['W_bz_' + g] for g in G4]), b_0=np.stack([P['b_0_' + g] for g in G4]),
            W_h=_stack(P, 'W_h_{}', G4), W_x=_stack(P, 'W_x_{}', G4))
        self.x = _tm(x); m = _tm(mask[:, :, None])[:, :, 0]
        hid, self.cache = hyper_scan_fwd(self.x, m, self.p, self.spec)
        return _tm(hid)

    def backward(self, dy):
        dx, g = hyper_scan_bwd(_tm(dy), self.x, self.cache, self.p, self.spec)
        G = {}; G4 = GATES_LHUC
        for nm in ('W_xhat', 'W_hhat', 'W_hhat_h', 'W_hhat_x', 'W_hhat_b', 'W_h', 'W_x'):
            _unstack_into(G, nm + '_{}', G4, g[nm])
        for nm in ('bhat', 'b_hhat_h', 'b_hhat_x'):
            _unstack_into(G, nm + '_{}', G4, g[nm], axis=0)
        for k, gn in enumerate(G4):
            for nm in ('W_hz', 'W_xz', 'W_bz', 'b_0'):
                G[nm + '_' + gn] = g[nm][k]
        self.grads = G
        return _tm(dx)


class HyperLHUCLSTMLayerO(OracleLayer):
    """libs/lasagne_libs/hyper_lhuc_layers.py:509-678."""
    def __init__(self, P, num_units, num_hyper_units, num_proj_units, grad_clipping=0, backwards=False,
                 reparam='relu'):
        self.P = P
        self.spec = dict(N=num_units, H=num_hyper_units, Z=num_proj_units, grad_clip=grad_clipping,
                         backwards=backwards, reparam=reparam)

    def forward(self, x, mask, cond=None):
        P = self.P; G4 = GATES_LHUC
        self.p = dict(W_xhat=_stack(P, 'W_xhat_{}', G4), bhat=_stack(P, 'bhat_{}', G4, 0),
                      W_hhat=_stack(P, 'W_hhat_{}', G4), W_hhat_h=P['W_hhat_h'], b_hhat_h=P['b_hhat_h'],
                      W_hz=P['W_hz'], W_h=_stack(P, 'W_h_{}', G4), W_x=_stack(P, 'W_x_{}', G4),
                      b=_stack(P, 'b_{}', G4, 0))
        self.x = _tm(x); m = _tm(mask[:, :, None])[:, :, 0]
        hid, self.cache = hyperlhuc_scan_fwd(self.x, m, self.p, self.spec)
        return _tm(hid)

    def backward(self, dy):
        dx, g = hyperlhuc_scan_bwd(_tm(dy), self.x, self.cache, self.p, self.spec)
        G = {}; G4 = GATES_LHUC
        for nm in ('W_xhat', 'W_hhat', 'W_h', 'W_x'):
            _unstack_into(G, nm + '_{}', G4, g[nm])
        for nm in ('bhat', 'b'):
            _unstack_into(G, nm + '_{}', G4, g[nm], axis=0)
        for nm in ('W_hhat_h', 'b_hhat_h', 'W_hz'):
            G[nm] = g[nm]
        self.grads = G
        return _tm(dx)


class ProjectionHyperLSTMLayerO(OracleLayer):
    """libs/lasagne_libs/hyper_layers.py:14-523.  forward returns [inner_hid | outer_hid] (:519); the
    builder feeds the outer half upward (models/gating_hyper_nets.py:150-171).  `d_inner` (B,T,F), when set
    before backward, is the gradient that a cost on the inner hidden sequence sends in (prj_hyper_aux_costs)."""
    def __init__(self, P, num_factors, num_units, grad_clipping=0, backwards=False, only_outer=False):
        self.P = P
        self.spec = dict(N=num_units, F=num_factors, grad_clip=grad_clipping, backwards=backwards)
        self.only_outer = only_outer      # the builder's SliceLayer(indices=slice(num_factors, None)) (:150-152)
        self.d_inner = None

    def forward(self, x, mask, cond=None):
        P = self.P; G = GATES_REC
        S = np.concatenate([_stack(P, 'W_inner_to_in_{}', G), _stack(P, 'W_inner_to_outer_{}', G),
                            _stack(P, 'W_inner_to_bias_{}', G)], axis=1)
        sb = np.concatenate([_stack(P, 'b_inner_to_in_{}', G, 0), _stack(P, 'b_inner_to_outer_{}', G, 0),
                             _stack(P, 'b_inner_to_bias_{}', G, 0)], axis=0)
        self.p = dict(Wi_x=_stack(P, 'W_in_to_inner_{}', G), Wi_i=_stack(P, 'W_inner_to_inner_{}', G),
                      Wi_o=_stack(P, 'W_outer_to_inner_{}', G), bi=_stack(P, 'b_inner_{}', G, 0),
                      Wi_p=P['W_inner_prj'], ai=P['W_inner_cell_ln'], bi_c=P['b_inner_cell_ln'],
                      Wo_x=_stack(P, 'W_in_to_outer_{}', G), Wo_o=_stack(P, 'W_outer_to_outer_{}', G),
                      S=S, s=sb, Wo_p=P['W_outer_prj'], ao=P['W_outer_cell_ln'], bo_c=P['b_outer_cell_ln'])
        self.x = _tm(x); m = _tm(mask[:, :, None])[:, :, 0]
        hi, ho, self.cache = prjhyper_scan_fwd(self.x, m, self.p, self.spec)
        self.inner_hid = _tm(hi)
        return _tm(ho) if self.only_outer else np.concatenate([_tm(hi), _tm(ho)], axis=-1)

    def backward(self, dy):
        F = self.spec['F']; N = self.spec['N']
        dyt = _tm(dy)
        if self.only_outer:
            di = np.zeros(dyt.shape[:2] + (F,), dyt.dtype) if self.d_inner is None else _tm(np.asarray(self.d_inner, dyt.dtype))
            dyt = np.concatenate([di, dyt], axis=-1)
        dx, g = prjhyper_scan_bwd(dyt[:, :, :F], dyt[:, :, F:], self.x, self.cache, self.p, self.spec)
        G = {}; G4 = GATES_REC
        for key, fmt in (('Wi_x', 'W_in_to_inner_{}'), ('Wi_i', 'W_inner_to_inner_{}'), ('Wi_o', 'W_outer_to_inner_{}'),
                         ('Wo_x', 'W_in_to_outer_{}'), ('Wo_o', 'W_outer_to_outer_{}')):
            _unstack_into(G, fmt, G4, g[key])
        _unstack_into(G, 'b_inner_{}', G4, g['bi'], axis=0)
        for k, nm in enumerate(('in', 'outer', 'bias')):
            _unstack_into(G, 'W_inner_to_%s_{}' % nm, G4, g['S'][:, k * 4 * N:(k + 1) * 4 * N])
            _unstack_into(G, 'b_inner_to_%s_{}' % nm, G4, g['s'][k * 4 * N:(k + 1) * 4 * N], axis=0)
        G['W_inner_prj'] = g['Wi_p']; G['W_inner_cell_ln'] = g['ai']; G['b_inner_cell_ln'] = g['bi_c']
        G['W_outer_prj'] = g['Wo_p']; G['W_outer_cell_ln'] = g['ao']; G['b_outer_cell_ln'] = g['bo_c']
        self.grads = G
        return _tm(dx)


class ScalingHyperLSTMLayerO(OracleLayer):
    """libs/lasagne_libs/hyper_layers.py:525-1205 with its options use_peepholes, use_layer_norm, learn_init.  forward
    returns the OUTER hidden sequence (what models/gating_hyper_nets.py:66-98 feeds upward); the inner hidden
    sequence is kept in self.inner_hid."""
    GI = ('W_inner_in_to_inner_{}', 'W_inner_hid_to_inner_{}', 'W_outer_hid_to_inner_{}', 'b_inner_{}')
    LN5 = GATES_REC + ('outcell',)

    def __init__(self, P, num_inner_units, num_outer_units, grad_clipping=0, backwards=False, use_peepholes=False,
                 use_layer_norm=False, learn_init=False):
        self.P = P
        self.spec = dict(N=num_outer_units, H=num_inner_units, grad_clip=grad_clipping, backwards=backwards,
                         peep=use_peepholes, ln=use_layer_norm, learn_init=learn_init)

    def forward(self, x, mask, cond=None):
        P = self.P; G = GATES_REC; sp = self.spec
        N, H = sp['N'], sp['H']
        U = sp.get('U', H)
        dt = np.asarray(x).dtype
        zH = np.zeros(U, dt); zHN = np.zeros((H, N), dt); zN = np.zeros(N, dt)
        pk = lambda fmt, z: np.concatenate([P[fmt.format(g)] if (sp['peep'] and g != 'cell') else z for g in G], axis=-1)
        lnv = lambda pre, wb, U: np.stack([P['%s_%s_ln_%s' % (wb, pre, g)] if sp['ln'] else np.zeros(U, dt) for g in self.LN5])
        ini = lambda nm, U: P[nm] if sp['learn_init'] else np.zeros((1, U), dt)
        self.p = dict(Wi_x=_stack(P, 'W_inner_in_to_inner_{}', G), Wi_h=_stack(P, 'W_inner_hid_to_inner_{}', G),
                      Wi_o=_stack(P, 'W_outer_hid_to_inner_{}', G), bi=_stack(P, 'b_inner_{}', G, 0),
                      Wo_x=_stack(P, 'W_outer_in_to_outer_{}', G), Wo_h=_stack(P, 'W_outer_hid_to_outer_{}', G),
                      S_in=_stack(P, 'W_inner_hid_to_outer_in_{}', G), s_in=_stack(P, 'b_inner_hid_to_outer_in_{}', G, 0),
                      S_hid=_stack(P, 'W_inner_hid_to_outer_hid_{}', G), s_hid=_stack(P, 'b_inner_hid_to_outer_hid_{}', G, 0),
                      S_bias=_stack(P, 'W_inner_hid_to_outer_bias_{}', G),
                      s_bias=_stack(P, 'b_inner_hid_to_outer_bias_{}', G, 0),
                      wpeep=pk('W_inner_cell_to_inner_{}', zH), S_cell=pk('W_inner_hid_to_outer_cell_{}', zHN),
                      s_cell=pk('b_inner_hid_to_outer_cell_{}', zN),
                      ln_ia=lnv('inner', 'W', U), ln_ib=lnv('inner', 'b', U), ln_oa=lnv('outer', 'W', N), ln_ob=lnv('outer', 'b', N),
                      ic0=ini('inner_cell_init', U), ih0=ini('inner_hid_init', H), oc0=ini('outer_cell_init', N),
                      oh0=ini('outer_hid_init', N))
        if 'U' in sp:
            self.p['P'] = P['W_inner_hid_prj']
        self.x = _tm(x); m = _tm(mask[:, :, None])[:, :, 0]
        hid_i, hid_o, self.cache = scaling_scan_fwd(self.x, m, self.p, self.spec)
        self.inner_hid = _tm(hid_i)
        return _tm(hid_o)

    def backward(self, dy, d_inner=None):
        dyo = _tm(dy)
        dyi = np.zeros(dyo.shape[:2] + (self.spec['H'],), dyo.dtype) if d_inner is None else _tm(d_inner)
        dx, g = scaling_scan_bwd(dyi, dyo, self.x, self.cache, self.p, self.spec)
        G = {}; G4 = GATES_REC; sp = self.spec
        for key, fmt in (('Wi_x', 'W_inner_in_to_inner_{}'), ('Wi_h', 'W_inner_hid_to_inner_{}'),
                         ('Wi_o', 'W_outer_hid_to_inner_{}'), ('Wo_x', 'W_outer_in_to_outer_{}'),
                         ('Wo_h', 'W_outer_hid_to_outer_{}'), ('S_in', 'W_inner_hid_to_outer_in_{}'),
                         ('S_hid', 'W_inner_hid_to_outer_hid_{}'), ('S_bias', 'W_inner_hid_to_outer_bias_{}')):
            _unstack_into(G, fmt, G4, g[key])
        for key, fmt in (('bi', 'b_inner_{}'), ('s_in', 'b_inner_hid_to_outer_in_{}'),
                         ('s_hid', 'b_inner_hid_to_outer_hid_{}'), ('s_bias', 'b_inner_hid_to_outer_bias_{}')):
            _unstack_into(G, fmt, G4, g[key], axis=0)
        if sp['peep']:
            tmp = {}
            _unstack_into(tmp, 'W_inner_cell_to_inner_{}', G4, g['wpeep'], axis=0)
            _unstack_into(tmp, 'W_inner_hid_to_outer_cell_{}', G4, g['S_cell'])
            _unstack_into(tmp, 'b_inner_hid_to_outer_cell_{}', G4, g['s_cell'], axis=0)
            G.update({k: v for k, v in tmp.items() if not k.endswith('_cell')})        # the cell gate has no peephole
        if sp['ln']:
            for pre, ka, kb in (('inner', 'ln_ia', 'ln_ib'), ('outer', 'ln_oa', 'ln_ob')):
                for k, gname in enumerate(self.LN5):
                    G['W_%s_ln_%s' % (pre, gname)] = g[ka][k]; G['b_%s_ln_%s' % (pre, gname)] = g[kb][k]
        if sp['learn_init']:
            G.update(inner_cell_init=g['ic0'], inner_hid_init=g['ih0'], outer_cell_init=g['oc0'], outer_hid_init=g['oh0'])
        if 'P' in g:
            G['W_inner_hid_prj'] = g['P']
        self.grads = G
        return _tm(dx)


class EqualScaleHyperLSTMLayerO(ScalingHyperLSTMLayerO):
    """libs/lasagne_libs/hyper_layers.py:1207-1890: ScalingHyperLSTMLayer's step (:1646-1779 is :963-1095 again) with an
    inner cell of num_outer_units columns per gate (slice_outer on the inner gates, :1660-1689) and
    inner_hid = (sigmoid(outgate) * tanh(outcell)) . W_inner_hid_prj (:1708-1709), num_inner_units wide, as the state that
    is carried, fed back (:1657) and mapped to the scales (:1714-1717)."""

    def __init__(self, P, num_inner_units, num_outer_units, grad_clipping=0, backwards=False, use_peepholes=False,
                 use_layer_norm=False, learn_init=False):
        ScalingHyperLSTMLayerO.__init__(self, P, num_inner_units, num_outer_units, grad_clipping, backwards, use_peepholes,
                                        use_layer_norm, learn_init)
        self.spec['U'] = num_outer_units


# --------------------------------------------------------------------------------------------
# Output layer + loss
# --------------------------------------------------------------------------------------------
def log_softmax(a):
    a = a - a.max(axis=-1, keepdims=True)                  # bin/forward_ln_lstmp.py:57-58
    return a - np.log(np.exp(a).sum(axis=-1, keepdims=True))


def philox4x32_10(counter, key):
    """Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11; the generator
    asr_dropout uses -- Theano's MRG31k3p stream of lasagne's DropoutLayer cannot be reproduced).
    counter: (..., 4) uint32, key: (2,) uint32 -> (..., 4) uint32."""
    c = [np.asarray(counter[..., i], np.uint64) for i in range(4)]
    k0, k1 = np.uint64(key[0]), np.uint64(key[1])
    M0, M1, W0, W1, MASK = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0x9E3779B9), np.uint64(0xBB67AE85), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0 = M0 * c[0]; p1 = M1 * c[2]
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c = [hi1 ^ c[1] ^ k0, lo1, hi0 ^ c[3] ^ k1, lo0]
        k0 = (k0 + W0) & MASK; k1 = (k1 + W1) & MASK
    return np.stack(c, axis=-1).astype(np.uint32)


def dropout_mask(rows, n, p, seed, stream_id):
    """The scaled mask keep/(1-p), (rows, n), that asr_dropout applies for (seed, stream_id): one Philox
    call per 4 consecutive columns of a row, counter = (row*ceil(n/4) + group, stream_id), keep <=>
    (word >> 8) * 2^-24 >= p (float32 comparison)."""
    ng = (n + 3) // 4
    idx = np.arange(rows * ng, dtype=np.uint64)
    ctr = np.stack([idx & np.uint64(0xFFFFFFFF), idx >> np.uint64(32),
                    np.full_like(idx, stream_id & 0xFFFFFFFF), np.full_like(idx, (stream_id >> 32) & 0xFFFFFFFF)], -1)
    w = philox4x32_10(ctr, (seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF))
    u = (w >> np.uint32(8)).astype(np.float32) * np.float32(2.0 ** -24)
    keep = (u >= np.float32(p)).reshape(rows, ng * 4)[:, :n]
    return keep.astype(np.float32) * (np.float32(1.0) / (np.float32(1.0) - np.float32(p)))


def softmax_ce(logits, target, target_mask):
    """libs/comp_graph_utils.py:29-45 (probabilities -> categorical_crossentropy) and
    scripts/hypernets/exp_prj_ln_lstm_fix.py:84-95 (logits -> log-softmax -> one-hot CE): the same
    number.  Returns ce_cost (= sum/n_batch, the differentiated scalar), ce_frame_sum, pred_idx,
    dlogits."""
    B, T, C = logits.shape
    lp = log_softmax(logits)
    ce = -np.take_along_axis(lp, target[:, :, None].astype(np.int64), axis=2)[:, :, 0] * target_mask
    ce_frame_sum = ce.sum()
    pred_idx = logits.argmax(axis=-1)
    dl = np.exp(lp)
    np.put_along_axis(dl, target[:, :, None].astype(np.int64),
                      np.take_along_axis(dl, target[:, :, None].astype(np.int64), 2) - 1.0, axis=2)
    dl = dl * (target_mask[:, :, None] / B)
    return ce_frame_sum / B, ce_frame_sum, pred_idx, dl


def adam_step(params, grads, m, v, t_prev, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """libs/lasagne_libs/updates.py:587-617.  In-place on lists of arrays; returns t."""
    t = t_prev + 1
    a_t = lr * np.sqrt(1 - b2 ** t) / (1 - b1 ** t)
    for p_, g_, m_, v_ in zip(params, grads, m, v):
        m_[...] = b1 * m_ + (1 - b1) * g_
        v_[...] = b2 * v_ + (1 - b2) * g_ ** 2
        p_[...] = p_ - a_t * m_ / (np.sqrt(v_) + eps)
    return t


def opt_step(kind, params, grads, s1, s2, lr, h1=0.0, h2=0.0, eps=0.0, t_prev=0):
    """The other updaters of libs/lasagne_libs/updates.py, in-place on lists of arrays; returns t.
    kind: 'sgd' :87-90, 'momentum' :143-147, 'nesterov' :244-249, 'adagrad' :357-361, 'rmsprop' :427-431,
    'adadelta' :515-529, 'adamax' :662-664,687-696."""
    t = t_prev + 1
    for i, (p_, g_) in enumerate(zip(params, grads)):
        if kind == 'sgd':
            p_[...] = p_ - lr * g_
        elif kind == 'momentum':
            x = h1 * s1[i] + (p_ - lr * g_) - p_
            s1[i][...] = x; p_[...] = p_ + x
        elif kind == 'nesterov':
            upd = p_ - lr * g_
            x = h1 * s1[i] + upd - p_
            s1[i][...] = x; p_[...] = h1 * x + upd
        elif kind == 'adagrad':
            s1[i][...] = s1[i] + g_ ** 2
            p_[...] = p_ - lr * g_ / np.sqrt(s1[i] + eps)
        elif kind == 'rmsprop':
            s1[i][...] = h1 * s1[i] + (1 - h1) * g_ ** 2
            p_[...] = p_ - lr * g_ / np.sqrt(s1[i] + eps)
        elif kind == 'adadelta':
            s1[i][...] = h1 * s1[i] + (1 - h1) * g_ ** 2
            u = g_ * np.sqrt(s2[i] + eps) / np.sqrt(s1[i] + eps)
            p_[...] = p_ - lr * u
            s2[i][...] = h1 * s2[i] + (1 - h1) * u ** 2
        elif kind == 'adamax':
            a_t = lr / (1 - h1 ** t)
            s1[i][...] = h1 * s1[i] + (1 - h1) * g_
            s2[i][...] = np.maximum(h2 * s2[i], np.abs(g_))
            p_[...] = p_ - a_t * s1[i] / (s2[i] + eps)
        else:
            raise KeyError(kind)
    return t


def total_norm_constraint(grads, max_norm, epsilon=1e-7):
    """libs/lasagne_libs/updates.py:848-855 -> (scaled grads, norm)."""
    norm = np.sqrt(sum(np.sum(g ** 2) for g in grads))
    mult = np.clip(norm, 0, max_norm) / (epsilon + norm)
    return [g * mult for g in grads], norm


def gen_win(batch, win_size, right_context=0):
    """libs/utils.py:178-191."""
    n_seq = batch[0].shape[1]
    for i in range(0, n_seq, win_size):
        yield tuple(a[:, i:i + win_size + right_context] for a in batch)


# --------------------------------------------------------------------------------------------
# Network = the wiring of libs/deep_lstm_builder.py:9-88, libs/deep_hyperlstm_builder.py:8-37,
# libs/deep_lhuclstm_builder.py:49-83,118-143, models/gating_hyper_nets.py:256-307,384-462,
# driven by the plain-data description the product's Network.describe() emits:
#   {'layers': [ {'fwd': {...}, 'bwd': {...}|None}, ... ], 'output': 'softmax'|'logits',
#    'ivector_concat': bool, 'cond': bool}
# each direction dict: {'kind', 'args': {...}, 'params': [names in add_param order]}
# --------------------------------------------------------------------------------------------
_KINDS = dict(lstm=LSTMLayerO, lhuc=LHUCLSTMLayerO, prj=ProjectLSTMLayerO,
              ln_prj=LayerNormProjectLSTMLayerO, icond_ln_prj=InputCondLayerNormProjectLSTMLayerO,
              hyper=HyperLSTMLayerO, rec=RecLayerO, scaling=ScalingHyperLSTMLayerO, equalscale=EqualScaleHyperLSTMLayerO,
              hyperlhuc=HyperLHUCLSTMLayerO, prjhyper=ProjectionHyperLSTMLayerO)


def prj_hyper_aux_costs(feats, mask, w_inter=0.0, w_intra=0.0):
    """scripts/hypernets/exp_prj_hyper.py:108-124: for each inner hidden sequence (B,T,F): the masked mean over
    time (:108-109), the variance over time around it with an UNMASKED sum of squares divided by the masked
    length (:112-113), the population variance of the means over the batch (:116-117);
    inter = -0.5 * sum_dirs mean(sample_var), intra = 0.5 * sum_dirs mean(seq_var) (:120-121).
    Returns (inter, intra, [d(w_inter*inter + w_intra*intra)/d feat for each feat])."""
    mask = np.asarray(mask, np.float64)
    m = mask[:, :, None]; L = mask.sum(1)[:, None]                     # (B,1)
    inter = 0.0; intra = 0.0; grads = []
    for x in feats:
        x = np.asarray(x, np.float64)
        B, T, F = x.shape
        mu = (x * m).sum(1) / L                                        # (B,F)
        dev = x - mu[:, None, :]
        seq_var = (dev ** 2).sum(1) / L
        sample_var = mu.var(0)
        inter += -0.5 * sample_var.mean(); intra += 0.5 * seq_var.mean()
        d_intra = (0.5 / (B * F)) * (2.0 * dev / L[:, :, None] - m * (2.0 / L[:, :, None] ** 2) * dev.sum(1)[:, None, :])
        d_inter = (-0.5 / F) * (2.0 * (mu - mu.mean(0)) / B)[:, None, :] * m / L[:, :, None]
        grads.append(w_intra * d_intra + w_inter * d_inter)
    return inter, intra, grads


class OracleNetwork(object):
    def __init__(self, desc, values, dtype=np.float64):
        self.desc = desc
        vals = [np.asarray(v, dtype=dtype) for v in values]
        self.dtype = dtype
        self.names = []
        self.layers = []
        k = 0
        self.seqsum = None
        if desc.get('seqsum'):          # libs/builder_utils.py:48-62: values come first (incomings-first order)
            n = len(desc['seqsum']['params'])
            self.seqsum = vals[:n]; k = n
        for L in desc['layers']:
            pair = []
            for d in ('fwd', 'bwd'):
                spec = L.get(d)
                if spec is None:
                    pair.append(None); continue
                P = {}
                for nm in spec['params']:
                    P[nm] = vals[k]; k += 1
                lay = _KINDS[spec['kind']](P, **spec['args'])
                lay.param_names = list(spec['params'])
                pair.append(lay)
            self.layers.append(pair)
        self.W = vals[k]; self.b = vals[k + 1]; k += 2
        assert k == len(vals), (k, len(vals))
        self.values = vals

    def tbptt_layers(self):
        return [p[0] for p in self.layers if getattr(p[0], 'stateful', False)]

    def forward(self, x, mask, cond=None, drop=None):
        """drop: None (deterministic) or the L+1 scaled masks keep/(1-p), (B,T,width) each, of the
        DropoutLayers in front of every layer and of the output layer (models/gating_hyper_nets.py:276-307;
        lasagne DropoutLayer: input * mask / (1-p))."""
        self.drop = None if drop is None else [np.asarray(d, self.dtype) for d in drop]
        x = np.asarray(x, self.dtype); mask = np.asarray(mask, self.dtype)
        if cond is not None:
            cond = np.asarray(cond, self.dtype)
        if self.desc.get('ivector_concat'):
            x = np.concatenate([x, cond], axis=-1)          # libs/deep_lstm_builder.py:14-16
        self.acts = [x]
        self.split = []
        if self.seqsum is not None:     # tanh MLP framewise, linear output, UNMASKED mean over time
            ss = self.seqsum; nl = len(ss) // 2 - 1
            a = [x]
            for i in range(nl):
                a.append(np.tanh(a[-1] @ ss[2 * i] + ss[2 * i + 1]))
            out = a[-1] @ ss[2 * nl] + ss[2 * nl + 1]
            self.ss_acts = a
            cond = out.mean(axis=1)[:, None, :]          # SeqSumLHUCLSTMLayer reads it as the embedding
        for l, (f, b) in enumerate(self.layers):
            xin = self.acts[-1] if self.drop is None else self.acts[-1] * self.drop[l]
            yf = f.forward(xin, mask, cond)
            self.split.append(yf.shape[-1])
            y = yf if b is None else np.concatenate([yf, b.forward(xin, mask, cond)], -1)
            self.acts.append(y)
        self.top = self.acts[-1] if self.drop is None else self.acts[-1] * self.drop[len(self.layers)]
        self.logits = self.top @ self.W + self.b           # libs/builder_utils.py:44-45
        return self.logits

    def posteriors(self, x, mask, cond=None):
        """What ff_fn returns: probabilities for build_* nets (libs/comp_graph_utils.py:321-333),
        logits for gating_hyper_nets models (models/gating_hyper_nets.py:304-306)."""
        lg = self.forward(x, mask, cond)
        return np.exp(log_softmax(lg)) if self.desc['output'] == 'softmax' else lg

    def loss_and_grads(self, x, mask, target, target_mask, cond=None, context=0, aux=None, drop=None):
        """aux = (w_inter, w_intra): adds the exp_prj_hyper.py costs on the inner hidden sequences of the first
        (ProjectionHyper) layer pair to the loss; self.aux_costs then holds (inter, intra)."""
        lg = self.forward(x, mask, cond, drop)
        for pair in self.layers:
            for l in pair:
                if l is not None and hasattr(l, 'd_inner'):
                    l.d_inner = None
        if aux is not None:
            f0, b0 = self.layers[0]
            inter, intra, dg = prj_hyper_aux_costs([f0.inner_hid, b0.inner_hid], mask, aux[0], aux[1])
            self.aux_costs = (inter, intra)
            f0.d_inner, b0.d_inner = dg
        tm = np.asarray(target_mask, self.dtype)
        if context:                                        # libs/comp_graph_utils.py:26-27,54-55
            tm = tm.copy(); tm[:, context:] = 0
        ce_cost, ce_sum, pred, dl = softmax_ce(lg, np.asarray(target), tm)
        B, T, C = lg.shape
        top = self.top
        gW = top.reshape(B * T, -1).T @ dl.reshape(B * T, C)
        gb = dl.sum((0, 1))
        dy = dl @ self.W.T
        if self.drop is not None:
            dy = dy * self.drop[len(self.layers)]
        grads_rev = []
        for li, ((f, b), nf) in enumerate(zip(reversed(self.layers), reversed(self.split))):
            if b is None:
                dx = f.backward(dy)
                grads_rev.append([f.grads[n] for n in f.param_names])
            else:
                dxf = f.backward(dy[..., :nf]); dxb = b.backward(dy[..., nf:])
                dx = dxf + dxb
                grads_rev.append([f.grads[n] for n in f.param_names] +
                                 [b.grads[n] for n in b.param_names])
            dy = dx
            if self.drop is not None:
                dy = dy * self.drop[len(self.layers) - 1 - li]
        grads = [g for lg_ in reversed(grads_rev) for g in lg_] + [gW, gb]
        if self.seqsum is not None:
            ss = self.seqsum; nl = len(ss) // 2 - 1
            demb = sum(l.demb for pair in self.layers for l in pair if l is not None)
            a = self.ss_acts
            Tn = a[0].shape[1]
            d = np.repeat(demb[:, None, :] / Tn, Tn, axis=1)
            sg = [None] * len(ss)
            for i in range(nl, -1, -1):
                sg[2 * i] = a[i].reshape(-1, a[i].shape[-1]).T @ d.reshape(-1, d.shape[-1])
                sg[2 * i + 1] = d.sum((0, 1))
                if i > 0:
                    d = (d @ ss[2 * i].T) * (1 - a[i] ** 2)
            grads = sg + grads
        if context:
            pred = pred[:, :context]
        return ce_cost, ce_sum, pred, grads

"""Parameter inventories of the reference layer classes: (name, shape, init) in `add_param` call
order, trainable parameters only (the non-trainable cell_init/hid_init are excluded exactly as
`lasagne.layers.get_all_params(network, trainable=True)` excludes them).  Pure python: used by the
layer classes, the pickle reader/writer and the tests.

Init kinds: 'orthogonal' | ('const', v) | ('normal', std) | 'glorot_uniform'.
"""

GATES_LHUC = ('ig', 'fg', 'c', 'og')                      # libs/lhuclstm_layers.py:53-56
GATES_REC = ('ingate', 'forgetgate', 'cell', 'outgate')   # libs/lasagne_libs/recurrent_layers.py:341-359


def lstm_params(D, N, P=0, use_layer_norm=False):
    """libs/lhuclstm_layers.py:32-41,52-71 (LSTMLayer / TBPTTLSTMLayer)."""
    K = P if P else N
    out = []
    for g in GATES_LHUC:
        out += [('W_h_' + g, (K, N), 'orthogonal'), ('W_x_' + g, (D, N), 'orthogonal'),
                ('b_' + g, (N,), ('const', 0.0))]
    if P:
        out.append(('W_p', (N, P), 'orthogonal'))
    if use_layer_norm:
        out += [('W_x_alpha', (4 * N,), ('const', 1.0)), ('W_h_alpha', (4 * N,), ('const', 1.0)),
                ('W_c_alpha', (N,), ('const', 1.0)), ('W_c_beta', (N,), ('const', 0.0))]
    return out


def lhuc_params(D, N, P, use_layer_norm, emb_dim, num_pred_units, num_pred_layers):
    """libs/lhuclstm_layers.py:327-348 appended to the LSTM parameters (:393-394)."""
    out = lstm_params(D, N, P, use_layer_norm)
    for i in range(num_pred_layers):
        ind = emb_dim if i == 0 else num_pred_units
        out += [('W_e_h_%d' % i, (ind, num_pred_units), ('normal', 0.01)),
                ('b_e_h_%d' % i, (num_pred_units,), ('const', 0.0))]
    ind = emb_dim if num_pred_layers == 0 else num_pred_units
    out += [('W_e_h_%d' % num_pred_layers, (ind, P if P else N), ('const', 0.0)),
            ('b_e_h_%d' % num_pred_layers, (P if P else N,), ('const', 1.0))]
    return out


def prj_params(D, N, F):
    """libs/lasagne_libs/recurrent_layers.py:329-377 (ProjectLSTMLayer, no peepholes)."""
    out = []
    for g in GATES_REC:
        out += [('W_in_to_' + g, (D, N), 'orthogonal'), ('W_hid_to_' + g, (F, N), 'orthogonal'),
                ('b_' + g, (N,), ('const', 1.0 if g == 'forgetgate' else 0.0))]
    out.append(('W_hid_prj', (N, F), 'orthogonal'))
    return out


def ln_prj_params(D, N, F):
    """libs/lasagne_libs/recurrent_layers.py:585-647 (LayerNormProjectLSTMLayer)."""
    out = prj_params(D, N, F)[:-1]
    out += [('W_alpha_in', (4 * N,), ('const', 1.0)), ('W_beta_in', (4 * N,), ('const', 0.0)),
            ('W_alpha_hid', (4 * N,), ('const', 1.0)), ('W_beta_hid', (4 * N,), ('const', 0.0)),
            ('W_alpha_cell', (N,), ('const', 1.0)), ('W_beta_cell', (N,), ('const', 0.0)),
            ('W_hid_prj', (N, F), 'orthogonal')]
    return out


def icond_ln_prj_params(D, N, F, I):
    """libs/lasagne_libs/recurrent_layers.py:1115-1188 (InputCondLayerNormProjectLSTMLayer)."""
    out = []
    for g in GATES_REC:
        out += [('W_in_to_' + g, (D, N), 'orthogonal'), ('W_hid_to_' + g, (N // 2, N), 'orthogonal')]
    out += [('W_hid_prj', (N, N // 2), 'orthogonal'),
            ('W_cond_prj', (I, F), 'orthogonal'), ('b_cond_prj', (F,), ('const', 0.0)),
            ('W_alpha_in', (F, 4 * N), ('const', 0.0)), ('b_alpha_in', (4 * N,), ('const', 1.0)),
            ('W_alpha_hid', (F, 4 * N), ('const', 0.0)), ('b_alpha_hid', (4 * N,), ('const', 1.0)),
            ('W_beta', (F, 4 * N), ('const', 0.0)), ('b_beta', (4 * N,), ('const', 0.0))]
    return out


def hyper_params(D, N, H, Z):
    """libs/lasagne_libs/hyper_lhuc_layers.py:251-276 (init_weights call order)."""
    out = []
    for g in GATES_LHUC:        # eq 10  (:253-256); Gate() default W_hid = Normal(0.1), b = 0
        out += [('W_hhat_' + g, (H, H), ('normal', 0.1)), ('W_xhat_' + g, (D + N, H), 'orthogonal'),
                ('bhat_' + g, (H,), ('const', 0.0))]
    for g in GATES_LHUC:        # eq 11  (:265-268)
        out += [('W_hhat_h_' + g, (H, Z), ('const', 0.0)), ('b_hhat_h_' + g, (Z,), ('const', 1.0)),
                ('W_hhat_x_' + g, (H, Z), ('const', 0.0)), ('b_hhat_x_' + g, (Z,), ('const', 1.0)),
                ('W_hhat_b_' + g, (H, Z), ('const', 0.0))]
    for g in GATES_LHUC:        # eq 12  (:271-274)
        out += [('W_hz_' + g, (Z, N), ('const', 1.0 / Z)), ('W_xz_' + g, (Z, N), ('const', 1.0 / Z)),
                ('W_bz_' + g, (Z, N), ('const', 0.0)), ('b_0_' + g, (N,), ('const', 0.0))]
    for g in GATES_LHUC:        # main LSTM (:240-243)
        out += [('W_h_' + g, (N, N), ('normal', 0.1)), ('W_x_' + g, (D, N), 'orthogonal')]
    return out


def dense_params(K, C):
    """Lasagne DenseLayer W (in,out), b (libs/builder_utils.py:44-45) and SequenceDenseLayer
    (libs/lasagne_libs/layers.py:77-82): GlorotUniform W, zero b."""
    return [('W', (K, C), 'glorot_uniform'), ('b', (C,), ('const', 0.0))]

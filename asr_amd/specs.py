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


# ------------------------------------------------------------------------------------------------
# The LSTM layer classes of libs/lasagne_libs/recurrent_layers.py as ONE parameterised family
# (SURVEY.md Appendix A).  Variant keys:
#   proj      None | 'factors' (num_factors or N/2, :322-324,:578-580) | 'half' (N/2, :856,:1120)
#             | 'prjs' (num_prjs, :2155)
#   ln        'none' | 'static' (ln / gate_ln) | 'sample' (sample_ln / sample_gate_ln)
#   per_gate  normalise each gate's N columns separately (gate_ln :17-25, sample_gate_ln :34-42)
#   ln_beta   static beta vectors on both LN terms (:711-712)
#   bias      per-gate b parameters (forget gate initialised to 1.0, :100,:605)
#   beta      for ln == 'sample': 'pre' = beta(B,4N) added to x.W BEFORE the LN (:968,:1233),
#             'post' = added after both LN terms (:2824,:4937); with `bias` the static b joins it (:2807)
#   cell_ln   outcell = ln(cell; alpha_cell, beta_cell) (:740,:2849)
#   cond      None | 'own' (pooled from the layer's own input, :944-946) | 'ext' (external i-vector input)
#   pool      'mean_tanh' (masked mean of tanh, :945-946) | 'max_relu' (unmasked max of relu, :1469-1470)
#   cinit     init of W_alpha_in / W_alpha_hid / W_beta: 0 or Uniform(0.01) (:2735-2753)
# ------------------------------------------------------------------------------------------------
def _v(proj=None, ln='none', per_gate=False, ln_beta=False, bias=True, beta=None, cell_ln=False, cond=None,
       pool='mean_tanh', cinit=0.0):
    return dict(proj=proj, ln=ln, per_gate=per_gate, ln_beta=ln_beta, bias=bias, beta=beta, cell_ln=cell_ln,
                cond=cond, pool=pool, cinit=cinit)


REC_CLASSES = {
    'LSTMLayer': _v(),                                                                   # :45
    'ProjectLSTMLayer': _v(proj='factors'),                                              # :290
    'LayerNormProjectLSTMLayer': _v(proj='factors', ln='static', ln_beta=True, cell_ln=True),   # :548
    'CondLayerNormProjectLSTMLayer': _v(proj='half', ln='sample', bias=False, beta='pre', cond='own'),      # :816
    'InputCondLayerNormProjectLSTMLayer': _v(proj='half', ln='sample', bias=False, beta='pre', cond='ext'),  # :1075
    'MaxCondLayerNormProjectLSTMLayer': _v(proj='half', ln='sample', bias=False, beta='pre', cond='own',
                                           pool='max_relu'),                              # :1340
    'MaxInputCondLayerNormProjectLSTMLayer': _v(proj='half', ln='sample', bias=False, beta='pre', cond='ext',
                                                pool='max_relu'),                         # :1599
    'ProjectLSTM_v0_1_Layer': _v(proj='prjs'),                                            # :1868
    'LayerNormProjectLSTM_v0_1_Layer': _v(proj='prjs', ln='static', cell_ln=True),        # :2125
    'LayerNormProjectLSTM_v0_2_Layer': _v(proj='prjs', ln='static', per_gate=True, cell_ln=True),   # :2380
    'CondLayerNormProjectLSTM_v0_1_Layer': _v(proj='prjs', ln='sample', beta='post', cell_ln=True, cond='own',
                                              cinit=0.01),                                # :2635
    'CondLayerNormProjectLSTM_v0_2_Layer': _v(proj='prjs', ln='sample', per_gate=True, beta='post', cell_ln=True,
                                              cond='own', cinit=0.01),                    # :2926
    'RegularLSTM_v0_1_Layer': _v(),                                                       # :3220
    'LayerNormRegularLSTM_v0_1_Layer': _v(ln='static', cell_ln=True),                     # :3467
    'LayerNormRegularLSTM_v0_2_Layer': _v(ln='static', per_gate=True, cell_ln=True),      # :3712
    'CondLayerNormRegularLSTM_v0_1_Layer': _v(ln='sample', beta='post', cell_ln=True, cond='own', cinit=0.01),   # :3957
    'CondLayerNormRegularLSTM_v0_2_Layer': _v(ln='sample', per_gate=True, beta='post', cell_ln=True, cond='own',
                                              cinit=0.01),                                # :4238
    'LayerNormLSTMLayer': _v(ln='static', ln_beta=True, cell_ln=True),                    # :4522
    'CondLayerNormLSTMLayer': _v(ln='sample', bias=False, beta='post', cond='own'),       # :4778
    'InputCondLayerNormLSTMLayer': _v(ln='sample', bias=False, beta='post', cond='ext'),  # :5034
}


def rec_out_width(cls, N, num_factors=None, num_prjs=None):
    """Width of the recurrent / output hidden vector (0 = no projection, i.e. N)."""
    proj = REC_CLASSES[cls]['proj']
    if proj is None:
        return 0
    if proj == 'factors':
        return num_factors or N // 2
    if proj == 'half':
        return N // 2
    return num_prjs


REC_PEEPHOLE_CLASSES = ('LSTMLayer', 'ProjectLSTMLayer', 'ProjectLSTM_v0_1_Layer', 'RegularLSTM_v0_1_Layer')


def rec_check_options(cls, peepholes=False, learn_init=False):
    """The classes that take the option at all (recurrent_layers.py:53,298,1876,3228 peepholes; learn_init: every
    class without conditioning, e.g. :52,:555; the conditioned classes keep constant zero initial rows, :909-918)."""
    if peepholes and cls not in REC_PEEPHOLE_CLASSES:
        raise ValueError('%s has no peepholes option' % cls)
    if learn_init and REC_CLASSES[cls]['cond']:
        raise ValueError('%s has no learn_init option' % cls)


def rec_params(cls, D, N, num_factors=None, num_prjs=None, cond_dim=0, peepholes=False, learn_init=False):
    """(name, shape, init) of the TRAINABLE parameters in add_param order for a class of REC_CLASSES.  The builders of
    models/gating_hyper_nets.py use peepholes=False, learn_init=False; with peepholes the three cell-to-gate vectors
    follow the gate parameters (:112-123), with learn_init the initial rows close the list (:125-135)."""
    rec_check_options(cls, peepholes, learn_init)
    v = REC_CLASSES[cls]
    P = rec_out_width(cls, N, num_factors, num_prjs)
    K = P if P else N
    F = num_factors
    out = []
    for g in GATES_REC:
        out += [('W_in_to_' + g, (D, N), 'orthogonal'), ('W_hid_to_' + g, (K, N), 'orthogonal')]
        if v['bias']:
            out.append(('b_' + g, (N,), ('const', 1.0 if g == 'forgetgate' else 0.0)))
    if peepholes:
        out += [('W_cell_to_' + g, (N,), ('uniform', 0.1)) for g in ('ingate', 'forgetgate', 'outgate')]
    if P and v['cond']:
        out.append(('W_hid_prj', (N, P), 'orthogonal'))
    if v['ln'] == 'static':
        out.append(('W_alpha_in', (4 * N,), ('const', 1.0)))
        if v['ln_beta']:
            out.append(('W_beta_in', (4 * N,), ('const', 0.0)))
        out.append(('W_alpha_hid', (4 * N,), ('const', 1.0)))
        if v['ln_beta']:
            out.append(('W_beta_hid', (4 * N,), ('const', 0.0)))
        if v['cell_ln']:
            out += [('W_alpha_cell', (N,), ('const', 1.0)), ('W_beta_cell', (N,), ('const', 0.0))]
    if v['cond']:
        cin = D if v['cond'] == 'own' else cond_dim
        wi = ('uniform', v['cinit']) if v['cinit'] else ('const', 0.0)
        out += [('W_cond_prj', (cin, F), 'orthogonal'), ('b_cond_prj', (F,), ('const', 0.0)),
                ('W_alpha_in', (F, 4 * N), wi), ('b_alpha_in', (4 * N,), ('const', 1.0)),
                ('W_alpha_hid', (F, 4 * N), wi), ('b_alpha_hid', (4 * N,), ('const', 1.0)),
                ('W_beta', (F, 4 * N), wi)]
        if not v['bias']:
            out.append(('b_beta', (4 * N,), ('const', 0.0)))
        if v['cell_ln']:      # the reference names the cell beta "num_factors" (:2762-2766, a typo kept
            out += [('W_alpha_cell', (N,), ('const', 1.0)), ('num_factors', (N,), ('const', 0.0))]   # for pickles)
    if P and not v['cond']:
        out.append(('W_hid_prj', (N, P), 'orthogonal'))
    if learn_init:
        out += [('cell_init', (1, N), ('const', 0.0)), ('hid_init', (1, K), ('const', 0.0))]
    return out


def scaling_hyper_params(D, H, N, use_peepholes=False, use_layer_norm=False, learn_init=False):
    """ScalingHyperLSTMLayer, libs/lasagne_libs/hyper_layers.py:570-834, TRAINABLE parameters in add_param order.
    Without an option its parameters are registered with trainable=False (:585-588,:602-609,:716-723,:736-743,
    :669-680,:823-834) and stay out of get_all_params(trainable=True): peephole vectors of the inner cell and the
    scale_outer_cell map (use_peepholes; never for the cell gate), the per-gate and out-cell layer-norm vectors
    (use_layer_norm), the four initial-state rows (learn_init).  The reference gives each scale bias the NAME of its
    weight ("W_inner_hid_to_outer_in_<gate>" twice, :694-700); names only label entries here, so the biases are
    called b_inner_hid_to_outer_* after the attributes they are bound to (:746-757)."""
    out = []
    for g in GATES_REC:
        fb = 1.0 if g == 'forgetgate' else 0.0
        out += [('W_inner_in_to_inner_' + g, (D, H), 'orthogonal'), ('W_inner_hid_to_inner_' + g, (H, H), 'orthogonal')]
        if use_peepholes and g != 'cell':
            out += [('W_inner_cell_to_inner_' + g, (H,), ('uniform', 0.1))]
        out += [('W_outer_hid_to_inner_' + g, (N, H), 'orthogonal'), ('b_inner_' + g, (H,), ('const', fb))]
        if use_layer_norm:
            out += [('W_inner_ln_' + g, (H,), ('const', 1.0)), ('b_inner_ln_' + g, (H,), ('const', 0.0))]
    if use_layer_norm:
        out += [('W_inner_ln_outcell', (H,), ('const', 1.0)), ('b_inner_ln_outcell', (H,), ('const', 0.0))]
    if learn_init:
        out += [('inner_cell_init', (1, H), ('const', 0.0)), ('inner_hid_init', (1, H), ('const', 0.0))]
    for g in GATES_REC:
        fb = 1.0 if g == 'forgetgate' else 0.0
        out += [('W_outer_in_to_outer_' + g, (D, N), 'orthogonal'),
                ('W_inner_hid_to_outer_in_' + g, (H, N), ('const', 0.0)), ('b_inner_hid_to_outer_in_' + g, (N,), ('const', 1.0)),
                ('W_outer_hid_to_outer_' + g, (N, N), 'orthogonal'),
                ('W_inner_hid_to_outer_hid_' + g, (H, N), ('const', 0.0)), ('b_inner_hid_to_outer_hid_' + g, (N,), ('const', 1.0))]
        if use_peepholes and g != 'cell':
            out += [('W_inner_hid_to_outer_cell_' + g, (H, N), 'orthogonal'), ('b_inner_hid_to_outer_cell_' + g, (N,), ('const', 1.0))]
        out += [('W_inner_hid_to_outer_bias_' + g, (H, N), 'orthogonal'), ('b_inner_hid_to_outer_bias_' + g, (N,), ('const', fb))]
        if use_layer_norm:
            out += [('W_outer_ln_' + g, (N,), ('const', 1.0)), ('b_outer_ln_' + g, (N,), ('const', 0.0))]
    if use_layer_norm:
        out += [('W_outer_ln_outcell', (N,), ('const', 1.0)), ('b_outer_ln_outcell', (N,), ('const', 0.0))]
    if learn_init:
        out += [('outer_cell_init', (1, N), ('const', 0.0)), ('outer_hid_init', (1, N), ('const', 0.0))]
    return out


def equal_scale_hyper_params(D, H, N, use_peepholes=False, use_layer_norm=False, learn_init=False):
    """EqualScaleHyperLSTMLayer, libs/lasagne_libs/hyper_layers.py:1252-1521, trainable parameters in add_param order: the
    ScalingHyperLSTMLayer with an inner cell as wide as the outer one (N gates columns per gate, :1258-1292) whose hidden
    state is projected to the H inner units by W_inner_hid_prj (:1340-1342, applied at :1709) before it is carried, fed
    back and mapped to the four scales.  Same trainable / non-trainable rules and naming as scaling_hyper_params."""
    out = []
    for g in GATES_REC:
        fb = 1.0 if g == 'forgetgate' else 0.0
        out += [('W_inner_in_to_inner_' + g, (D, N), 'orthogonal'), ('W_inner_hid_to_inner_' + g, (H, N), 'orthogonal')]
        if use_peepholes and g != 'cell':
            out += [('W_inner_cell_to_inner_' + g, (N,), ('uniform', 0.1))]
        out += [('W_outer_hid_to_inner_' + g, (N, N), 'orthogonal'), ('b_inner_' + g, (N,), ('const', fb))]
        if use_layer_norm:
            out += [('W_inner_ln_' + g, (N,), ('const', 1.0)), ('b_inner_ln_' + g, (N,), ('const', 0.0))]
    out += [('W_inner_hid_prj', (N, H), 'orthogonal')]
    if use_layer_norm:
        out += [('W_inner_ln_outcell', (N,), ('const', 1.0)), ('b_inner_ln_outcell', (N,), ('const', 0.0))]
    if learn_init:
        out += [('inner_cell_init', (1, N), ('const', 0.0)), ('inner_hid_init', (1, H), ('const', 0.0))]
    inner = [r[0] for r in scaling_hyper_params(D, H, N, use_peepholes, use_layer_norm, learn_init)]
    outer_from = inner.index('W_outer_in_to_outer_ingate')
    return out + scaling_hyper_params(D, H, N, use_peepholes, use_layer_norm, learn_init)[outer_from:]


def hyperlhuc_params(D, N, H, Z, reparam='relu'):
    """HyperLHUCLSTMLayer, libs/lasagne_libs/hyper_lhuc_layers.py:533-555 (init_weights, then
    init_main_lstm_weights :521-525).  Gate.b defaults to Constant(0.) (lasagne Gate)."""
    out = []
    for g in GATES_LHUC:
        out += [('W_hhat_' + g, (H, H), ('normal', 0.1)), ('W_xhat_' + g, (N + D, H), 'orthogonal'),
                ('bhat_' + g, (H,), ('const', 0.0))]
    out += [('W_hhat_h', (H, Z), ('const', 0.0)), ('b_hhat_h', (Z,), ('const', 1.0)),
            ('W_hz', (Z, N), ('const', 1.0 / Z))]
    for g in GATES_LHUC:
        out += [('W_h_' + g, (N, N), ('normal', 0.1)), ('W_x_' + g, (D, N), 'orthogonal'), ('b_' + g, (N,), ('const', 0.0))]
    return out


def prjhyper_params(D, F, N):
    """ProjectionHyperLSTMLayer, libs/lasagne_libs/hyper_layers.py:55-208, trainable parameters in
    add_param order.  The in-scale bias is registered under the NAME "b_outer_to_in_<gate>" (:135) but
    bound to b_inner_to_in_<gate> (:160); the attribute name is used here.  Forget biases are 0 (the gate
    helpers are called without bias_const, :81-99,164-194)."""
    out = []
    for g in GATES_REC:
        out += [('W_in_to_inner_' + g, (D, N), 'orthogonal'), ('W_inner_to_inner_' + g, (F, N), 'orthogonal'),
                ('W_outer_to_inner_' + g, (F, N), 'orthogonal'), ('b_inner_' + g, (N,), ('const', 0.0))]
    out += [('W_inner_prj', (N, F), 'orthogonal'), ('W_inner_cell_ln', (N,), ('const', 1.0)),
            ('b_inner_cell_ln', (N,), ('const', 0.0))]
    for g in GATES_REC:
        out += [('W_in_to_outer_' + g, (D, N), 'orthogonal'), ('W_outer_to_outer_' + g, (F, N), 'orthogonal'),
                ('W_inner_to_in_' + g, (F, N), ('const', 0.0)), ('b_inner_to_in_' + g, (N,), ('const', 1.0)),
                ('W_inner_to_outer_' + g, (F, N), ('const', 0.0)), ('b_inner_to_outer_' + g, (N,), ('const', 1.0)),
                ('W_inner_to_bias_' + g, (F, N), 'orthogonal'), ('b_inner_to_bias_' + g, (N,), ('const', 0.0))]
    out += [('W_outer_prj', (N, F), 'orthogonal'), ('W_outer_cell_ln', (N,), ('const', 1.0)),
            ('b_outer_cell_ln', (N,), ('const', 0.0))]
    return out

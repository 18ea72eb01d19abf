"""models/gating_hyper_nets.py builders on the hot path (same names / keyword names):
  deep_projection_ln_lstm_model        models/gating_hyper_nets.py:256-307
  deep_projection_ivector_ln_model     models/gating_hyper_nets.py:384-462
  deep_projection_lstm_model           plain ProjectLSTMLayer stack (same wiring, no LN)
Every model is bidirectional, each layer and the output layer preceded by DropoutLayer(p=dropout)
(reference default 0.2; active in training_fn = deterministic=False, identity in predict / ff functions:
engine.Network._dropout_fwd), output = SequenceDenseLayer(nonlinearity=None) -> logits."""
from . import specs
from .engine import Network, LstmPair


def _check_dropout(dropout):
    p = float(dropout or 0.0)
    if not 0.0 <= p < 1.0:
        raise ValueError('dropout must be in [0, 1), got %r' % (dropout,))
    return p


def deep_projection_ln_lstm_model(input_var, mask_var, num_inputs, num_outputs, num_layers, num_factors,
                                  num_units, grad_clipping=1, dropout=0.2, device=None, seed=0):
    net = Network(device or 'cuda:0')
    net.dropout = _check_dropout(dropout)
    F = num_factors or num_units // 2          # recurrent_layers.py:578-580
    D = num_inputs
    for l in range(num_layers):
        pair = LstmPair(net, len(net.pairs), 'ln_prj', D, num_units, F, True, grad_clipping, 2)
        net.add_pair(pair, [specs.ln_prj_params(D, num_units, F)] * 2)
        D = pair.out_dim
    net.set_output(D, num_outputs, 'logits')
    return net.finalize(seed)


def deep_projection_lstm_model(input_var, mask_var, num_inputs, num_outputs, num_layers, num_factors,
                               num_units, grad_clipping=1, dropout=0.2, device=None, seed=0):
    net = Network(device or 'cuda:0')
    net.dropout = _check_dropout(dropout)
    F = num_factors or num_units // 2
    D = num_inputs
    for l in range(num_layers):
        pair = LstmPair(net, len(net.pairs), 'prj', D, num_units, F, False, grad_clipping, 2)
        net.add_pair(pair, [specs.prj_params(D, num_units, F)] * 2)
        D = pair.out_dim
    net.set_output(D, num_outputs, 'logits')
    return net.finalize(seed)


def deep_projection_ivector_ln_model(input_var, cond_var, mask_var, num_inputs, num_outputs, num_layers,
                                     num_conds, num_factors, num_units, grad_clipping=1, dropout=0.2,
                                     cond_dim=100, device=None, seed=0):
    """-> (output_layer, cond_layer_list).  The first `num_conds` layers are
    InputCondLayerNormProjectLSTMLayer (hidden width num_units/2, recurrent_layers.py:1120,1141),
    the rest ProjectLSTMLayer(num_factors) (models/gating_hyper_nets.py:413-449).  The i-vector input
    layer is hard-wired to 100 dims in the reference (:400); `cond_dim` exposes it."""
    net = Network(device or 'cuda:0')
    net.dropout = _check_dropout(dropout)
    net.uses_cond = True
    D = num_inputs
    cond_layers = []
    for l in range(num_layers):
        if l < num_conds:
            pair = LstmPair(net, len(net.pairs), 'icond_ln_prj', D, num_units, num_units // 2, False,
                            grad_clipping, 2, I=cond_dim, F=num_factors)
            net.add_pair(pair, [specs.icond_ln_prj_params(D, num_units, num_factors, cond_dim)] * 2)
            cond_layers.append(pair)
        else:
            pair = LstmPair(net, len(net.pairs), 'prj', D, num_units, num_factors, False, grad_clipping, 2)
            net.add_pair(pair, [specs.prj_params(D, num_units, num_factors)] * 2)
        D = pair.out_dim
    net.set_output(D, num_outputs, 'logits')
    return net.finalize(seed), cond_layers


# ------------------------------------------------------------------------------------------------
# The remaining LSTM builders of models/gating_hyper_nets.py on the generic recurrent_layers.py
# family (asr_amd/engine_rec.py).  Same names / argument order / defaults.
class CondLayerHandle(object):
    """What the reference appends to cond_layer_list: one entry per direction with get_sample_feat()
    (libs/lasagne_libs/recurrent_layers.py:1337-1338)."""
    def __init__(self, pair, d):
        self.pair = pair; self.d = d

    def get_sample_feat(self):
        return self.pair.get_sample_feat(self.d)


def _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout, plain_cls,
               cond_cls=None, num_conds=0, num_factors=None, num_prjs=None, cond_dim=0, is_bidir=True,
               device=None, seed=0, training_dropout_ok=False, rec_options=None):
    """rec_options: peepholes / learn_init / only_return_hidden of the plain_cls layers (recurrent_layers.py:52-57; no
    builder of models/gating_hyper_nets.py sets them).  only_return_final yields one row per utterance and does not
    chain into this frame-level network: use RecPair directly."""
    from .engine_rec import RecPair
    rec_options = dict(rec_options or {})
    if rec_options.get('only_return_final'):
        raise ValueError('only_return_final does not chain into a frame-level network; call RecPair directly')
    net = Network(device or 'cuda:0')
    net.dropout = _check_dropout(dropout)
    ndir = 2 if is_bidir else 1
    if cond_cls and specs.REC_CLASSES[cond_cls]['cond'] == 'ext':
        net.uses_cond = True
    D = num_inputs
    cond_layers = []
    for l in range(num_layers):
        cls = cond_cls if (cond_cls and l < num_conds) else plain_cls
        pair = RecPair(net, len(net.pairs), cls, D, num_units, grad_clipping, ndir, num_factors=num_factors,
                       num_prjs=num_prjs, cond_dim=cond_dim, **(rec_options if cls == plain_cls else {}))
        net.add_pair(pair, [pair.spec_per_dir()] * ndir)
        if cls == cond_cls:
            cond_layers += [CondLayerHandle(pair, d) for d in range(ndir)]
        D = pair.out_dim
    net.set_output(D, num_outputs, 'logits')
    net.finalize(seed)
    return net, cond_layers


def deep_projection_cond_ln_model(input_var, mask_var, num_inputs, num_outputs, num_layers, num_conds,
                                  num_factors, num_units, grad_clipping=1, dropout=0.2, device=None, seed=0):
    """models/gating_hyper_nets.py:309-382 -> (output_layer, cond_layer_list)."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout, 'ProjectLSTMLayer',
                      'CondLayerNormProjectLSTMLayer', num_conds, num_factors, device=device, seed=seed)


def deep_projection_max_cond_ln_model(input_var, mask_var, num_inputs, num_outputs, num_layers, num_conds,
                                      num_factors, num_units, grad_clipping=1, dropout=0.2, device=None, seed=0):
    """:544-617."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout, 'ProjectLSTMLayer',
                      'MaxCondLayerNormProjectLSTMLayer', num_conds, num_factors, device=device, seed=seed)


def deep_projection_max_ivector_ln_model(input_var, cond_var, mask_var, num_inputs, num_outputs, num_layers,
                                         num_conds, num_factors, num_units, grad_clipping=1, dropout=0.2,
                                         cond_dim=100, device=None, seed=0):
    """:694-772 (i-vector input hard-wired to 100 dims at :710; `cond_dim` exposes it)."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout, 'ProjectLSTMLayer',
                      'MaxInputCondLayerNormProjectLSTMLayer', num_conds, num_factors, cond_dim=cond_dim,
                      device=device, seed=seed)


def deep_lstm_model(input_var, mask_var, num_inputs, num_outputs, num_layers, num_units, grad_clipping=1,
                    dropout=0.2, device=None, seed=0):
    """:858-906."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout, 'LSTMLayer',
                      device=device, seed=seed)[0]


def deep_ln_lstm_model(input_var, mask_var, num_inputs, num_outputs, num_layers, num_units, grad_clipping=1,
                       dropout=0.2, device=None, seed=0):
    """:908-956."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout, 'LayerNormLSTMLayer',
                      device=device, seed=seed)[0]


def deep_cond_ln_model(input_var, mask_var, num_inputs, num_outputs, num_layers, num_factors, num_conds,
                       num_units, grad_clipping=1, dropout=0.2, device=None, seed=0):
    """:958-1029 (note the reference's argument order: num_factors BEFORE num_conds)."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout, 'LSTMLayer',
                      'CondLayerNormLSTMLayer', num_conds, num_factors, device=device, seed=seed)


def deep_ivector_ln_model(input_var, cond_var, mask_var, num_inputs, num_outputs, num_layers, num_conds,
                          num_factors, num_units, grad_clipping=1, dropout=0.2, cond_dim=100, device=None, seed=0):
    """:1031-1107."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout, 'LSTMLayer',
                      'InputCondLayerNormLSTMLayer', num_conds, num_factors, cond_dim=cond_dim, device=device,
                      seed=seed)


def deep_prj_lstm_model_v1(input_var, mask_var, num_inputs, num_outputs, num_layers, num_units, num_prjs,
                           grad_clipping=1, dropout=0.2, device=None, seed=0):
    """:1113-1164."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout,
                      'ProjectLSTM_v0_1_Layer', num_prjs=num_prjs, device=device, seed=seed)[0]


def deep_prj_lstm_ln_model_v1(input_var, mask_var, num_inputs, num_outputs, num_layers, num_units, num_prjs,
                              grad_clipping=1, dropout=0.2, device=None, seed=0):
    """:1166-1217."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout,
                      'LayerNormProjectLSTM_v0_1_Layer', num_prjs=num_prjs, device=device, seed=seed)[0]


def deep_prj_lstm_ln_model_v2(input_var, mask_var, num_inputs, num_outputs, num_layers, num_units, num_prjs,
                              grad_clipping=1, dropout=0.2, device=None, seed=0):
    """:1219-1270 (per-gate layer norm)."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout,
                      'LayerNormProjectLSTM_v0_2_Layer', num_prjs=num_prjs, device=device, seed=seed)[0]


def _dln(num_layers, num_conds):
    return num_layers if num_conds == -1 else num_conds


def deep_prj_lstm_dln_model_v1(input_var, mask_var, num_inputs, num_outputs, num_units, num_prjs, num_factors,
                               num_layers, num_conds=-1, grad_clipping=0.0, dropout=0.2, device=None, seed=0):
    """:1272-1352."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout,
                      'LayerNormProjectLSTM_v0_1_Layer', 'CondLayerNormProjectLSTM_v0_1_Layer',
                      _dln(num_layers, num_conds), num_factors, num_prjs, device=device, seed=seed)


def deep_prj_lstm_dln_model_v2(input_var, mask_var, num_inputs, num_outputs, num_units, num_prjs, num_factors,
                               num_layers, num_conds=-1, grad_clipping=0.0, dropout=0.2, device=None, seed=0):
    """:1354-1436."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout,
                      'LayerNormProjectLSTM_v0_2_Layer', 'CondLayerNormProjectLSTM_v0_2_Layer',
                      _dln(num_layers, num_conds), num_factors, num_prjs, device=device, seed=seed)


def deep_reg_lstm_model_v1(input_var, mask_var, num_inputs, num_outputs, num_layers, num_units, grad_clipping=1,
                           dropout=0.2, device=None, seed=0):
    """:1438-1486."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout,
                      'RegularLSTM_v0_1_Layer', device=device, seed=seed)[0]


def deep_reg_lstm_ln_model_v1(input_var, mask_var, num_inputs, num_outputs, num_layers, num_units,
                              grad_clipping=1, dropout=0.2, device=None, seed=0):
    """:1488-1536."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout,
                      'LayerNormRegularLSTM_v0_1_Layer', device=device, seed=seed)[0]


def deep_reg_lstm_ln_model_v2(input_var, mask_var, num_inputs, num_outputs, num_layers, num_units,
                              grad_clipping=1, dropout=0.2, device=None, seed=0):
    """:1538-1586 (per-gate layer norm)."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout,
                      'LayerNormRegularLSTM_v0_2_Layer', device=device, seed=seed)[0]


def deep_reg_lstm_dln_model_v1(input_var, mask_var, num_inputs, num_outputs, num_units, num_factors, num_layers,
                               num_conds=-1, grad_clipping=0.0, dropout=0.2, device=None, seed=0):
    """:1588-1663."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout,
                      'LayerNormRegularLSTM_v0_1_Layer', 'CondLayerNormRegularLSTM_v0_1_Layer',
                      _dln(num_layers, num_conds), num_factors, device=device, seed=seed)


def deep_reg_lstm_dln_model_v2(input_var, mask_var, num_inputs, num_outputs, num_units, num_factors, num_layers,
                               num_conds=-1, grad_clipping=0.0, dropout=0.2, device=None, seed=0):
    """:1665-1743."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout,
                      'LayerNormRegularLSTM_v0_2_Layer', 'CondLayerNormRegularLSTM_v0_2_Layer',
                      _dln(num_layers, num_conds), num_factors, device=device, seed=seed)


def uni_lstm_model(input_var, mask_var, num_inputs, num_outputs, num_layers, num_units, grad_clipping=1,
                   dropout=0.2, device=None, seed=0):
    """:1745-1784 (the only unidirectional builder of the file)."""
    return _build_rec(num_inputs, num_outputs, num_layers, num_units, grad_clipping, dropout,
                      'RegularLSTM_v0_1_Layer', is_bidir=False, device=device, seed=seed)[0]


def deep_scaling_hyper_model(input_var, mask_var, num_inputs, num_outputs, num_inner_units_list,
                             num_outer_units_list, use_peepholes=False, use_layer_norm=False, learn_init=False,
                             grad_clipping=1.0, get_inner_hid=False, use_softmax=True, device=None, seed=0):
    """models/gating_hyper_nets.py:27-115: bidirectional ScalingHyperLSTMLayer stack; each layer's
    OUTER hidden sequences (fwd | bwd) feed the next layer and the SequenceDenseLayer.  With
    get_inner_hid=True the reference returns the inner-hidden slice layers before the output layer;
    here that is a list of (pair, direction) handles with .get_output() -> (T*B, inner) of the last
    forward, followed by the Network.  The reference defaults (no option) run on the persistent HyperLSTM scan;
    use_peepholes / use_layer_norm / learn_init (scripts/timit/exp_scalehyper.py uses the last two) run on the
    step-loop path of asr_amd/csrc/step_scaling.hip."""
    from .engine_scaling import ScalingPair, ScalingStepPair
    options = bool(use_peepholes or use_layer_norm or learn_init)
    net = Network(device or 'cuda:0')
    D = num_inputs
    inner = []
    for H, N in zip(num_inner_units_list, num_outer_units_list):
        if options:        # per-gate layer norms do not fit the persistent scan: the step-loop path (step_scaling.hip)
            pair = ScalingStepPair(net, len(net.pairs), D, N, H, grad_clipping or 0, 2, use_peepholes, use_layer_norm,
                                   learn_init)
        else:
            pair = ScalingPair(net, len(net.pairs), D, N, H, grad_clipping or 0, 2)
        net.add_pair(pair, [specs.scaling_hyper_params(D, H, N, use_peepholes, use_layer_norm, learn_init)] * 2)
        inner += [InnerHidHandle(pair, d) for d in range(2)]
        D = pair.out_dim
    net.set_output(D, num_outputs, 'softmax' if use_softmax else 'logits')
    net.finalize(seed)
    return inner + [net] if get_inner_hid else net


def deep_equal_scale_hyper_model(input_var, mask_var, num_inputs, num_outputs, num_inner_units_list,
                                 num_outer_units_list, use_peepholes=False, use_layer_norm=False, learn_init=False,
                                 grad_clipping=1.0, get_inner_hid=False, use_softmax=True, device=None, seed=0):
    """deep_scaling_hyper_model's wiring (models/gating_hyper_nets.py:27-115) with EqualScaleHyperLSTMLayer
    (libs/lasagne_libs/hyper_layers.py:1207-1890) in place of ScalingHyperLSTMLayer.  The reference defines the layer but
    no model function or script instantiates it; this builder exists so that the layer has a caller with the same
    signature as its sibling's."""
    from .engine_scaling import EqualScaleStepPair
    net = Network(device or 'cuda:0')
    D = num_inputs
    inner = []
    for H, N in zip(num_inner_units_list, num_outer_units_list):
        pair = EqualScaleStepPair(net, len(net.pairs), D, N, H, grad_clipping or 0, 2, use_peepholes, use_layer_norm, learn_init)
        net.add_pair(pair, [specs.equal_scale_hyper_params(D, H, N, use_peepholes, use_layer_norm, learn_init)] * 2)
        inner += [InnerHidHandle(pair, d) for d in range(2)]
        D = pair.out_dim
    net.set_output(D, num_outputs, 'softmax' if use_softmax else 'logits')
    net.finalize(seed)
    return inner + [net] if get_inner_hid else net


class InnerHidHandle(object):
    def __init__(self, pair, d):
        self.pair = pair; self.d = d

    def get_output(self):
        return self.pair.inner_hid(self.d)


def deep_projection_hyper_model(input_var, mask_var, num_inputs, num_outputs, num_layers, num_factors, num_units,
                                grad_clipping=1, device=None, seed=0):
    """models/gating_hyper_nets.py:117-203 -> [output_layer, fwd_inner, bwd_inner]: the first layer is a
    bidirectional ProjectionHyperLSTMLayer whose OUTER hidden sequences feed the stack, the remaining
    layers are ProjectLSTMLayer(num_factors); logits output.  The inner entries have .get_output() ->
    (T*B, num_factors) of the last forward."""
    from .engine_prjhyper import PrjHyperPair
    from .engine_rec import RecPair
    net = Network(device or 'cuda:0')
    D = num_inputs
    inner = []
    for l in range(num_layers):
        if l == 0:
            pair = PrjHyperPair(net, len(net.pairs), D, num_units, num_factors, grad_clipping or 0, 2)
            net.add_pair(pair, [specs.prjhyper_params(D, num_factors, num_units)] * 2)
            inner = [InnerHidHandle(pair, d) for d in range(2)]
        else:
            pair = RecPair(net, len(net.pairs), 'ProjectLSTMLayer', D, num_units, grad_clipping, 2,
                           num_factors=num_factors)
            net.add_pair(pair, [pair.spec_per_dir()] * 2)
        D = pair.out_dim
    net.set_output(D, num_outputs, 'logits')
    net.finalize(seed)
    return [net] + inner

"""Builder functions with the reference's names, argument order and return arity.

  build_deep_lstm / build_deep_lstm_tbptt   libs/deep_lstm_builder.py:9-42, :44-88
  build_deep_hyperlstm                      libs/deep_hyperlstm_builder.py:8-37
  build_deep_lhuclstm_ivector (+ _tbptt)    libs/deep_lhuclstm_builder.py:49-83, :118-143

`input_var` / `mask_var` / `ivector_var` were Theano symbols in the reference; here they are only
tested for truthiness (a non-None `ivector_var` switches the i-vector input on), data is passed to
the compiled callables of asr_amd.comp_graph_utils.  Each builder returns an asr_amd.engine.Network
(the stand-in for the Lasagne output layer) with parameters initialised as the reference does.
"""
from . import specs
from .engine import Network, LstmPair


def _new(device, seed):
    return Network(device or 'cuda:0')


def _stack_lstm(net, kind, input_dim, num_layers, num_units, num_proj_units, grad_clipping, is_bidir,
                use_layer_norm, stateful=False, context=0, backward_on_top=False, emb_dim=0,
                num_pred_units=0, num_pred_layers=0):
    D = input_dim
    ndir_default = 2 if is_bidir else 1
    for li in range(1, num_layers + 1):
        ndir = ndir_default
        if backward_on_top and li < num_layers:        # libs/deep_lstm_builder.py:68-70
            ndir = 1
        pair = LstmPair(net, len(net.pairs), kind, D, num_units, num_proj_units, use_layer_norm,
                        grad_clipping, ndir, I=emb_dim, npu=num_pred_units, npl=num_pred_layers,
                        stateful=stateful, context=context)
        if kind == 'lhuc':
            sp = specs.lhuc_params(D, num_units, num_proj_units, use_layer_norm, emb_dim, num_pred_units,
                                   num_pred_layers)
        else:
            sp = specs.lstm_params(D, num_units, num_proj_units, use_layer_norm)
        net.add_pair(pair, [sp] * ndir)
        D = pair.out_dim
    return D


def build_deep_lstm(input_var, mask_var, input_dim, num_layers, num_units, num_proj_units, output_dim,
                    grad_clipping, is_bidir, use_layer_norm, ivector_dim, ivector_var=None,
                    device=None, seed=0):
    net = _new(device, seed)
    D = input_dim
    if ivector_var is not None and ivector_var is not False:
        net.ivector_concat = True; net.uses_cond = True
        D += ivector_dim
    top = _stack_lstm(net, 'lstm', D, num_layers, num_units, num_proj_units, grad_clipping, is_bidir,
                      use_layer_norm)
    net.set_output(top, output_dim, 'softmax')
    return net.finalize(seed)


def build_deep_lstm_tbptt(input_var, mask_var, input_dim, num_layers, num_units, num_proj_units,
                          output_dim, batch_size, context, grad_clipping, is_bidir, use_layer_norm,
                          ivector_dim, ivector_var=None, backward_on_top=False, device=None, seed=0):
    net = _new(device, seed)
    D = input_dim
    if ivector_var is not None and ivector_var is not False:
        net.ivector_concat = True; net.uses_cond = True
        D += ivector_dim
    top = _stack_lstm(net, 'lstm', D, num_layers, num_units, num_proj_units, grad_clipping, is_bidir,
                      use_layer_norm, stateful=True, context=context, backward_on_top=backward_on_top)
    net.set_output(top, output_dim, 'softmax')
    net.finalize(seed)
    tb = net.tbptt_layers()
    for l in tb:
        l.reset(batch_size)
    return net, tb


def build_deep_hyperlstm(layer_name, input_var, mask_var, input_dim, num_layers, num_units,
                         num_hyper_units, num_proj_units, output_dim, grad_clipping, is_bidir,
                         ivector_dim, ivector_var=None, device=None, seed=0):
    from .engine_hyper import HyperPair
    net = _new(device, seed)
    D = input_dim
    if ivector_var is not None and ivector_var is not False:
        net.ivector_concat = True; net.uses_cond = True
        D += ivector_dim
    ndir = 2 if is_bidir else 1
    for li in range(num_layers):
        pair = HyperPair(net, len(net.pairs), D, num_units, num_hyper_units, num_proj_units,
                         grad_clipping, ndir)
        net.add_pair(pair, [specs.hyper_params(D, num_units, num_hyper_units, num_proj_units)] * ndir)
        D = pair.out_dim
    net.set_output(D, output_dim, 'softmax')
    return net.finalize(seed)


def build_deep_hyperlhuclstm(layer_name, input_var, mask_var, input_dim, num_layers, num_units,
                             num_hyper_units, num_proj_units, output_dim, grad_clipping, is_bidir, ivector_dim,
                             ivector_var=None, reparam='relu', device=None, seed=0):
    """The wiring of build_deep_hyperlstm (libs/deep_hyperlstm_builder.py:8-37) with
    HyperLHUCLSTMLayer (libs/lasagne_libs/hyper_lhuc_layers.py:509-678) as the layer class.  The
    reference defines the class but none of its builders instantiates it (its `layer_name` argument is
    ignored, :20-27); this builder exists so that the class is reachable the same way."""
    from .engine_hyper import HyperLHUCPair
    net = _new(device, seed)
    D = input_dim
    if ivector_var is not None and ivector_var is not False:
        net.ivector_concat = True; net.uses_cond = True
        D += ivector_dim
    ndir = 2 if is_bidir else 1
    for li in range(num_layers):
        pair = HyperLHUCPair(net, len(net.pairs), D, num_units, num_hyper_units, num_proj_units,
                             grad_clipping, ndir, reparam=reparam)
        net.add_pair(pair, [specs.hyperlhuc_params(D, num_units, num_hyper_units, num_proj_units)] * ndir)
        D = pair.out_dim
    net.set_output(D, output_dim, 'softmax')
    return net.finalize(seed)


def _build_lhuc_ivector(input_dim, num_layers, num_units, num_proj_units, output_dim, num_pred_layers,
                        num_pred_units, batch_size, grad_clipping, is_bidir, use_layer_norm,
                        use_ivector_input, ivector_dim, stateful, device, seed):
    net = _new(device, seed)
    net.uses_cond = True
    D = input_dim
    if use_ivector_input:
        net.ivector_concat = True
        D += ivector_dim
    top = _stack_lstm(net, 'lhuc', D, num_layers, num_units, num_proj_units, grad_clipping, is_bidir,
                      use_layer_norm, stateful=stateful, emb_dim=ivector_dim,
                      num_pred_units=num_pred_units, num_pred_layers=num_pred_layers)
    net.set_output(top, output_dim, 'softmax')
    net.finalize(seed)
    return net


def _build_lhuc_seqsum(input_dim, num_layers, num_units, num_proj_units, output_dim, num_pred_layers,
                       num_pred_units, num_seqsum_units, num_seqsum_layers, seqsum_output_dim, grad_clipping,
                       is_bidir, use_layer_norm, use_ivector_input, ivector_dim, stateful, device, seed):
    from .engine import SeqSumNet
    net = _new(device, seed)
    D = input_dim
    if use_ivector_input:
        net.ivector_concat = True; net.uses_cond = True
        D += ivector_dim
    net.seqsum = SeqSumNet(net, D, num_seqsum_units, num_seqsum_layers, seqsum_output_dim)
    top = _stack_lstm(net, 'lhuc', D, num_layers, num_units, num_proj_units, grad_clipping, is_bidir,
                      use_layer_norm, stateful=stateful, emb_dim=seqsum_output_dim,
                      num_pred_units=num_pred_units, num_pred_layers=num_pred_layers)
    net.set_output(top, output_dim, 'softmax')
    net.finalize(seed)
    return net


def build_deep_lhuclstm_seqsum(layer_name, input_var, mask_var, input_dim, num_layers, num_units,
                               num_proj_units, output_dim, num_pred_layers, num_pred_units, num_seqsum_units,
                               num_seqsum_layers, seqsum_output_dim, batch_size, grad_clipping, is_bidir,
                               use_layer_norm, use_ivector_input, ivector_dim, ivector_var=None, device=None,
                               seed=0):
    """libs/deep_lhuclstm_builder.py:8-45 -> (network, speaker_layer).  The speaker embedding is the
    sequence-summarising network's output (libs/builder_utils.py:48-62), shared by all layers."""
    net = _build_lhuc_seqsum(input_dim, num_layers, num_units, num_proj_units, output_dim, num_pred_layers,
                             num_pred_units, num_seqsum_units, num_seqsum_layers, seqsum_output_dim,
                             grad_clipping, is_bidir, use_layer_norm, use_ivector_input, ivector_dim, False,
                             device, seed)
    return net, net.seqsum


def build_deep_uni_lhuclstm_seqsum_tbptt(layer_name, input_var, mask_var, input_dim, num_layers, num_units,
                                         num_proj_units, output_dim, num_pred_layers, num_pred_units,
                                         num_seqsum_units, num_seqsum_layers, seqsum_output_dim, batch_size,
                                         grad_clipping, use_layer_norm, use_ivector_input, ivector_dim,
                                         ivector_var=None, device=None, seed=0):
    """libs/deep_lhuclstm_builder.py:85-116 -> (network, speaker_layer, tbptt_layers); the summary is
    taken over each window the network is fed."""
    net = _build_lhuc_seqsum(input_dim, num_layers, num_units, num_proj_units, output_dim, num_pred_layers,
                             num_pred_units, num_seqsum_units, num_seqsum_layers, seqsum_output_dim,
                             grad_clipping, False, use_layer_norm, use_ivector_input, ivector_dim, True,
                             device, seed)
    tb = net.tbptt_layers()
    for l in tb:
        l.reset(batch_size)
    return net, net.seqsum, tb


def build_deep_lhuclstm_ivector(layer_name, input_var, mask_var, input_dim, num_layers, num_units,
                                num_proj_units, output_dim, num_pred_layers, num_pred_units,
                                num_seqsum_units, num_seqsum_layers, seqsum_output_dim, batch_size,
                                grad_clipping, is_bidir, use_layer_norm, use_ivector_input, ivector_dim,
                                ivector_var, device=None, seed=0):
    """Returns (network, speaker_layer); speaker_layer is the i-vector input (None here: data is fed
    to the compiled callables).  The reference passes `batch_size` into the layer constructor where it
    collides with `backwards=` (libs/deep_lhuclstm_builder.py:65-68); we follow the
    documented signature and give `batch_size` to the stateful TBPTT variant only."""
    net = _build_lhuc_ivector(input_dim, num_layers, num_units, num_proj_units, output_dim,
                              num_pred_layers, num_pred_units, batch_size, grad_clipping, is_bidir,
                              use_layer_norm, use_ivector_input, ivector_dim, False, device, seed)
    return net, None


def build_deep_uni_lhuclstm_ivector_tbptt(layer_name, input_var, mask_var, input_dim, num_layers,
                                          num_units, num_proj_units, output_dim, num_pred_layers,
                                          num_pred_units, num_seqsum_units, num_seqsum_layers,
                                          seqsum_output_dim, batch_size, grad_clipping, use_layer_norm,
                                          use_ivector_input, ivector_dim, ivector_var, device=None,
                                          seed=0):
    """libs/deep_lhuclstm_builder.py:118-143 -> (network, speaker_layer, tbptt_layers)."""
    net = _build_lhuc_ivector(input_dim, num_layers, num_units, num_proj_units, output_dim,
                              num_pred_layers, num_pred_units, batch_size, grad_clipping, False,
                              use_layer_norm, use_ivector_input, ivector_dim, True, device, seed)
    tb = net.tbptt_layers()
    for l in tb:
        l.reset(batch_size)
    return net, None, tb

"""The training / prediction functions that the scripts around models/gating_hyper_nets.py build for
themselves (scripts/dln/ted/exp_prj_lstm_dln.py:88-200; the same block is in scripts/hypernets/*.py and
scripts/wsj/*.py): logits -> log-softmax CE, optional total_norm_constraint, `updater` chosen by name.

  build_trainer(...)   -> (training_fn, train_lr, updater_params)
      training_fn(x, mask, y, ymask) -> [frame_loss, frame_accr, train_feat_loss, network_grads_norm]
  build_predictor(...) -> predict_fn(x, mask, y, ymask) -> [frame_prd_idx, frame_loss]
  eval_network(predict_fn, data_stream) -> (nll, fer)                                   (:186-215)

Symbolic arguments are placeholders, as in asr_amd.comp_graph_utils.  The sample-feature variance
regulariser (`feat_reg`, :120-131) enters the backward pass of the conditioning layers as one more GEMM
with the constant centring matrix.  L2 weight decay
(`weight_decay > 0`: `apply_penalty(network_reg_params, l2) * 10**(-weight_decay)`, :129-131) adds
2*10**(-weight_decay)*W to the gradients of the regularizable parameters before the norm constraint."""
import numpy as np
import torch

from .comp_graph_utils import _dev, _reducer
from .updates import SharedScalar, total_norm_constraint


_GATES = ('ingate', 'forgetgate', 'cell', 'outgate')


def lasagne_regularizable(name):
    """The `regularizable` tag of the parameters of libs/lasagne_libs/recurrent_layers.py and of
    SequenceDenseLayer: add_param's default is True; False for the gate biases b_<gate> (:90,339,595),
    b_cond_prj, b_beta, the LN betas W_beta_in / W_beta_hid / W_beta_cell (:624-659) and the cell beta the
    v0_x cond classes register under the name "num_factors"; DenseLayer's b is False as well."""
    if name in ('b', 'b_cond_prj', 'b_beta', 'num_factors'):
        return False
    if name.startswith('W_beta'):
        return False
    if name.startswith('b_') and name[2:] in _GATES:
        return False
    return True


def build_trainer(input_data, input_mask, target_data, target_mask, network_params, network_reg_params,
                  output_layer, cond_layer_list, feat_reg, weight_decay, updater, learning_rate,
                  max_grad_norm=0.0, load_updater_params=None):
    network = output_layer
    if feat_reg and cond_layer_list:
        for cl in cond_layer_list:             # train_feat_loss*feat_reg, averaged over the cond layers (:128)
            if not hasattr(cl, 'pair') or not hasattr(cl.pair, '_cond_bwd'):
                raise NotImplementedError('asr_amd: feat_reg needs the cond layers of asr_amd.engine_rec')
            cl.pair.feat_reg = float(feat_reg) / len(cond_layer_list)
    wd_coef = 10.0 ** (-weight_decay) if (weight_decay and weight_decay > 0) else 0.0
    grads = None
    if max_grad_norm and max_grad_norm > 0.:
        grads, _ = total_norm_constraint(tensor_vars=None, max_norm=max_grad_norm, return_norm=True)
    train_lr = SharedScalar(learning_rate, 'train_lr')
    update, updater_params = updater(loss_or_grads=grads, params=network_params, learning_rate=train_lr,
                                     load_params_dict=load_updater_params)
    red = _reducer(network)
    f32 = torch.float32

    def training_fn(x, mask, y, ymask):
        ym = _dev(network, ymask, f32)
        ce, ok, _ = network.loss(_dev(network, x, f32), _dev(network, mask, f32), _dev(network, y, torch.int32),
                                 ym, cond=None, train=True)
        network.backward()
        red.finish()
        red.agree_on_status()
        if wd_coef:
            network.add_weight_decay(wd_coef, lasagne_regularizable)
        gn = update.apply(network, want_sqrt=True)
        out = torch.cat([ce, ok, gn, ym.sum().reshape(1)]).cpu().numpy()
        nfr = max(float(out[3]), 1.0)
        # train_feat_loss (:120-127) is reported even at feat_reg = 0: -mean(var(sample_feat, axis=0))
        # averaged over the conditioning layers; a host-side read-out of the pooled features
        fl = 0.0
        for cl in cond_layer_list or []:
            f = cl.get_sample_feat()
            f = f.detach().cpu().numpy() if hasattr(f, 'detach') else np.asarray(f)
            fl += -float(np.mean(np.var(f.astype(np.float64), axis=0)))
        if cond_layer_list:
            fl /= len(cond_layer_list)
        return [np.float32(out[0] / nfr), np.float32(out[1] / nfr), np.float32(fl), np.float32(out[2])]

    return training_fn, train_lr, updater_params


def build_prj_hyper_trainer(input_data, input_mask, target_data, target_mask, network_params, output_layer,
                            fwd_inner_layer, bwd_inner_layer, updater, inter_weight, intra_weight, learning_rate,
                            load_updater_params=None):
    """scripts/hypernets/exp_prj_hyper.py:70-146: CE/B + inter_weight*inter + intra_weight*intra, where the two
    costs are the batch / time variances of the inner hidden sequences of the ProjectionHyper layer
    (asr_amd.engine_prjhyper.PrjHyperPair.aux_costs).  Returns (training_fn, updater_params) as the reference
    does; training_fn -> [frame_loss, frame_accr, grads_norm, inter_cost, intra_cost]."""
    network = output_layer
    pair = fwd_inner_layer.pair
    if bwd_inner_layer.pair is not pair:
        raise ValueError('asr_amd: the two inner layers must come from one deep_projection_hyper_model call')
    train_lr = SharedScalar(learning_rate, 'train_lr')
    update, updater_params = updater(loss_or_grads=None, params=network_params, learning_rate=train_lr,
                                     load_params_dict=load_updater_params)
    red = _reducer(network)
    f32 = torch.float32

    def training_fn(x, mask, y, ymask):
        ym = _dev(network, ymask, f32)
        ce, ok, _ = network.loss(_dev(network, x, f32), _dev(network, mask, f32), _dev(network, y, torch.int32),
                                 ym, cond=None, train=True)
        costs = pair.aux_costs(float(inter_weight), float(intra_weight))       # device tensor (inter, intra)
        network.backward()
        red.finish()
        red.agree_on_status()
        gn = update.apply(network, want_sqrt=True)
        out = torch.cat([ce, ok, gn, ym.sum().reshape(1), costs]).cpu().numpy()
        nfr = max(float(out[3]), 1.0)
        return [np.float32(out[0] / nfr), np.float32(out[1] / nfr), np.float32(out[2]), np.float32(out[4]),
                np.float32(out[5])]

    return training_fn, updater_params


def build_predictor(input_data, input_mask, target_data, target_mask, output_layer):
    network = output_layer
    f32 = torch.float32

    def predict_fn(x, mask, y, ymask):
        ym = _dev(network, ymask, f32)
        ce, ok, pred = network.loss(_dev(network, x, f32), _dev(network, mask, f32),
                                    _dev(network, y, torch.int32), ym, cond=None, train=False)
        out = torch.cat([ce, ym.sum().reshape(1)]).cpu().numpy()
        network.check_status()
        return [pred.cpu().numpy(), np.float32(out[0] / max(float(out[1]), 1.0))]

    return predict_fn


def eval_network(predict_fn, data_stream):
    """scripts/dln/ted/exp_prj_lstm_dln.py:186-215: batch-averaged NLL and frame error rate."""
    total_nll = 0.; total_fer = 0.; cnt = 0
    for cnt, data in enumerate(data_stream.get_epoch_iterator(), start=1):
        x = np.asarray(data[0], np.float32); m = np.asarray(data[1], np.float32)
        y = np.asarray(data[2]); ym = np.asarray(data[3], np.float32)
        pred, nll = predict_fn(x, m, y, ym)
        match = (np.asarray(pred).reshape(y.shape) == y) * ym
        total_nll += float(nll)
        total_fer += 1.0 - float(match.sum() / ym.sum())
    cnt = max(cnt, 1)
    return total_nll / cnt, total_fer / cnt

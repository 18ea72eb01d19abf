"""Host-side utilities with the reference's names and behaviour:
  save_network / load_or_init_model / find_reload_model   libs/utils.py:80-85, :154-175
  set_model_param_value / get_model_param_values          libs/lasagne_libs/utils.py:2-17
  gen_win / gen_win_test (TBPTT windowing)                libs/utils.py:178-205
  StopWatch, show_status, EvalRecord history helpers      libs/utils.py:29-50, :87-107

Pickle layout: [ [ndarray, ...], updater_vals, epoch_cnt ].  The reference's
get_update_params_values has no `return` (libs/param_utils.py:20-23), so element [1] of every
reference pickle is None; we write None as well unless an optimiser state dict is passed explicitly,
and accept either on load.  Python-2 pickles are read with encoding='latin1'; ours are written with
protocol 2 so the reference's Python 2 can read them back.
"""
from __future__ import print_function

import os
import pickle
import time
from collections import namedtuple
from operator import attrgetter

import numpy as np

EvalRecord = namedtuple('EvalRecord', ['train_ce_frame', 'valid_ce_frame', 'valid_fer', 'test_ce_frame',
                                       'test_fer'])   # scripts/exp_deep_lstm2.py:100-101


def _param_list(network_params):
    """Accepts the Network itself or the list returned by Network.get_all_params()."""
    if hasattr(network_params, 'get_all_params'):
        return network_params.get_all_params(trainable=True)
    return list(network_params)


def get_model_param_values(params_list):
    return [p.get_value() for p in _param_list(params_list)]


def set_model_param_value(params_var_list, params_val_list):
    params_var_list = _param_list(params_var_list)
    if len(params_var_list) != len(params_val_list):
        raise ValueError("mismatch: got %d values to set %d parameters" %
                         (len(params_val_list), len(params_var_list)))
    shapes = [p.get_value().shape for p in params_var_list]
    for shp, v in zip(shapes, params_val_list):
        if tuple(shp) != tuple(np.shape(v)):
            raise ValueError("mismatch: parameter has shape %r but value to "
                             "set has shape %r" % (tuple(shp), tuple(np.shape(v))))
    for p, v in zip(params_var_list, params_val_list):
        p.set_value(v)


def get_update_params_values(params_var_dict):
    """The reference returns None here (missing `return`); keep that so pickles are byte-compatible
    in structure.  Use `trainer_params` objects' own state_dict() for a full optimiser resume."""
    return None


def save_network(network_params, trainer_params, epoch_cnt, save_path):
    with open(save_path, 'wb') as f:
        cur_network_params_val = get_model_param_values(network_params)
        cur_trainer_params_val = get_update_params_values(trainer_params)
        pickle.dump([cur_network_params_val, cur_trainer_params_val, epoch_cnt], f, protocol=2)


def load_network_pickle(path):
    with open(path, 'rb') as f:
        try:
            obj = pickle.load(f)
        except UnicodeDecodeError:
            f.seek(0)
            obj = pickle.load(f, encoding='latin1')
    if not isinstance(obj, (list, tuple)) or len(obj) != 3:
        raise ValueError('%s is not a [params, updater_vals, epoch] model pickle' % path)
    return obj


def load_or_init_model(network_params, args):
    pretrain_update_params_val = None
    pretrain_total_epoch_cnt = 0
    if getattr(args, 'reload_model', None):
        print('Loading the model: {}'.format(args.reload_model))
        pretrain_network_params_val, pretrain_update_params_val, pretrain_total_epoch_cnt = \
            load_network_pickle(args.reload_model)
        set_model_param_value(network_params, pretrain_network_params_val)
    return pretrain_update_params_val, pretrain_total_epoch_cnt


def find_reload_model(args):
    if not args.reload_model and not args.no_reload:
        reload_path = args.save_path + '_last_model.pkl'
        if os.path.exists(reload_path):
            print('Previously trained model detected: {}'.format(reload_path))
            print('Training continues')
            args.reload_model = reload_path


def save_eval_history(eval_history, save_path):
    with open(save_path, 'wb') as f:
        pickle.dump([tuple(r) for r in eval_history], f, protocol=2)


def best_fer(eval_history):
    acopy = list(eval_history)
    acopy.sort(key=attrgetter('valid_fer'))
    return acopy[0].valid_fer


def show_status(save_path, ce_frame, network_grads_norm, batch_idx, batch_size, epoch_idx):
    model = save_path.split('/')[-1]
    print('--')
    print('Model Name: {} (Epoch {})'.format(model, epoch_idx))
    print('Train CE {} (batch {}, {} examples so far): '.format(ce_frame, batch_idx, batch_idx * batch_size))
    print('Gradient Norm: {}'.format(network_grads_norm))


class StopWatch(object):
    """libs/utils.py:29-50."""
    def __init__(self):
        self.start_time = time.time()

    def reset(self):
        self.start_time = time.time()

    def elapsed(self):
        return time.time() - self.start_time

    def print_elapsed(self):
        time_diff = time.time() - self.start_time
        print('Took {:.1f}s ({:.1f}m; {:.1f}h)'.format(time_diff, time_diff / 60, time_diff / 3600))


def gen_win(batch, win_size, right_context=0):
    """Windows of `win_size` frames plus `right_context` look-ahead frames that the next window
    re-processes (libs/utils.py:178-191)."""
    input_data, input_mask, ivector_data, ivector_mask, target_data, target_mask = batch
    n_batch, n_seq, n_feat = input_data.shape
    for i in range(0, n_seq, win_size):
        from_idx = i
        to_idx = i + win_size + right_context
        yield (input_data[:, from_idx:to_idx, :], input_mask[:, from_idx:to_idx],
               ivector_data[:, from_idx:to_idx, :] if ivector_data is not None else None,
               ivector_mask[:, from_idx:to_idx] if ivector_mask is not None else None,
               target_data[:, from_idx:to_idx], target_mask[:, from_idx:to_idx])


def gen_win_test(batch, win_size, right_context=0):
    """libs/utils.py:193-205."""
    input_data, input_mask, ivector_data, ivector_mask = batch
    n_batch, n_seq, n_feat = input_data.shape
    for i in range(0, n_seq, win_size):
        from_idx = i
        to_idx = i + win_size + right_context
        yield (input_data[:, from_idx:to_idx, :], input_mask[:, from_idx:to_idx],
               ivector_data[:, from_idx:to_idx, :] if ivector_data is not None else None,
               ivector_mask[:, from_idx:to_idx] if ivector_mask is not None else None)

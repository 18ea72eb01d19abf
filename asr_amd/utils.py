"""Host-side bookkeeping around the engine.  The contract taken from the reference is the DATA and the
call surface, not the code:

  * the model file (libs/utils.py:80-85, :154-166): a pickle of [ [ndarray, ...], updater_vals, epoch ]
    with the arrays in get_all_params(trainable=True) order; `updater_vals` is None in every file the
    reference writes (its get_update_params_values has no `return`, libs/param_utils.py:20-23), and we
    write None as well.  Written with protocol 2 so Python 2 can read it back; read with a RESTRICTED
    unpickler (numpy arrays / scalars and plain containers only) -- a model file named on a command
    line must not be able to run code;
  * the evaluation history file (libs/utils.py:87-99): a pickle of a list of 5-tuples
    (train_ce_frame, valid_ce_frame, valid_fer, test_ce_frame, test_fer);
  * the TBPTT windows (libs/utils.py:178-205): frames [i, i + win + right_context) for i = 0, win, ...;
  * the status lines the training scripts print (libs/utils.py:102-107, :46-50).

The function names the reference's scripts import (save_network, load_or_init_model, find_reload_model,
gen_win, gen_win_test, StopWatch, show_status, best_fer, save_eval_history, EvalRecord) are thin fronts
over `Checkpoint`, `EvalHistory` and `windows()` below.
"""
from __future__ import print_function

import io
import os
import pickle
import time
from collections import OrderedDict, namedtuple

import numpy as np

EvalRecord = namedtuple('EvalRecord', ['train_ce_frame', 'valid_ce_frame', 'valid_fer', 'test_ce_frame',
                                       'test_fer'])   # scripts/exp_deep_lstm2.py:100-101


# ------------------------------------------------------------------------------------------------
# restricted unpickling
def _np_core(sub):
    """numpy's core sub-module under the name this numpy has (numpy._core from 2.0 on, numpy.core before)."""
    import importlib
    try:
        return importlib.import_module('numpy._core.' + sub)
    except ImportError:
        return importlib.import_module('numpy.core.' + sub)


def _latin1_bytes(text, encoding):
    """What protocol-2 pickles of Python-3 bytes objects call: codecs.encode(str, 'latin1').  Only that form is
    accepted (the encoding name of a general codecs.encode drives a codec lookup that imports modules)."""
    if not isinstance(text, str) or encoding not in ('latin1', 'latin-1'):
        raise pickle.UnpicklingError('asr_amd refuses codecs.encode(%s, %r)' % (type(text).__name__, encoding))
    return text.encode('latin1')


def _numpy_global(module, name):
    mod = module.replace('numpy.core', 'numpy._core')      # numpy < 2 wrote numpy.core.*
    if mod == 'numpy._core.multiarray' and name in ('_reconstruct', 'scalar'):
        return getattr(_np_core('multiarray'), name)
    if mod == 'numpy._core.numeric' and name == '_frombuffer':
        return _np_core('numeric')._frombuffer
    if mod == 'numpy' and name in ('ndarray', 'dtype'):
        return getattr(np, name)
    return None


class _DataUnpickler(pickle.Unpickler):
    """Resolves only what an array list needs: numpy's array / scalar / dtype reconstructors,
    OrderedDict (an updater-state dict) and this module's EvalRecord.  Anything else -- os.system,
    builtins.eval, a class with __reduce__ -- raises UnpicklingError before it can be called."""
    def find_class(self, module, name):
        g = _numpy_global(module, name)
        if g is not None:
            return g
        if (module, name) == ('collections', 'OrderedDict'):
            return OrderedDict
        if name == 'EvalRecord' and module in ('asr_amd.utils', 'libs.utils', 'utils', '__main__'):
            return EvalRecord                   # this module's, or the reference's libs/utils.py:87-89 namedtuple
        if (module, name) in (('_codecs', 'encode'), ('codecs', 'encode')):
            return _latin1_bytes                # protocol-2 bytes objects written by Python 3
        raise pickle.UnpicklingError('asr_amd refuses to unpickle %s.%s: model and history files hold '
                                     'numpy arrays and plain containers only' % (module, name))


def _read_data_pickle(path):
    with open(path, 'rb') as f:
        raw = f.read()
    try:
        return _DataUnpickler(io.BytesIO(raw)).load()
    except UnicodeDecodeError:                  # a Python 2 str holding array bytes
        return _DataUnpickler(io.BytesIO(raw), encoding='latin1').load()


# ------------------------------------------------------------------------------------------------
# parameters <-> value lists
def _handles(network_or_params):
    """The Network itself or the list its get_all_params() returns (objects with get_value/set_value)."""
    if hasattr(network_or_params, 'get_all_params'):
        return network_or_params.get_all_params(trainable=True)
    return list(network_or_params)


def get_model_param_values(params_list):
    return [h.get_value() for h in _handles(params_list)]


def set_model_param_value(params_var_list, params_val_list):
    """All-or-nothing: every count / shape is checked before the first value is written
    (messages of libs/lasagne_libs/utils.py:2-14)."""
    handles = _handles(params_var_list)
    values = list(params_val_list)
    if len(handles) != len(values):
        raise ValueError("mismatch: got %d values to set %d parameters" % (len(values), len(handles)))
    for h, v in zip(handles, values):
        have, want = tuple(np.shape(v)), tuple(h.get_value().shape)
        if have != want:
            raise ValueError("mismatch: parameter has shape %r but value to set has shape %r" % (want, have))
    for h, v in zip(handles, values):
        h.set_value(v)


def get_update_params_values(params_var_dict):
    """None, as in every file the reference writes (see the module docstring)."""
    return None


class Checkpoint(object):
    """One model file."""
    def __init__(self, values, updater_values=None, epoch=0):
        self.values = list(values)
        self.updater_values = updater_values
        self.epoch = int(epoch)

    @classmethod
    def of(cls, network_or_params, epoch, updater_values=None):
        return cls(get_model_param_values(network_or_params), updater_values, epoch)

    @classmethod
    def read(cls, path):
        obj = _read_data_pickle(path)
        if not isinstance(obj, (list, tuple)) or len(obj) != 3 or not isinstance(obj[0], (list, tuple)):
            raise ValueError('%s is not a [params, updater_vals, epoch] model pickle' % path)
        return cls([np.asarray(a) for a in obj[0]], obj[1], obj[2])

    def write(self, path):
        tmp = path + '.tmp%d' % os.getpid()         # a killed run never leaves half a model behind
        with open(tmp, 'wb') as f:
            pickle.dump([self.values, self.updater_values, self.epoch], f, protocol=2)
        os.replace(tmp, path)

    def as_list(self):
        return [self.values, self.updater_values, self.epoch]


def save_network(network_params, trainer_params, epoch_cnt, save_path):
    Checkpoint.of(network_params, epoch_cnt, get_update_params_values(trainer_params)).write(save_path)


def load_network_pickle(path):
    return Checkpoint.read(path).as_list()


def load_or_init_model(network_params, args):
    """-> (updater values or None, number of epochs already trained)."""
    path = getattr(args, 'reload_model', None)
    if not path:
        return None, 0
    print('Loading the model: {}'.format(path))
    ck = Checkpoint.read(path)
    set_model_param_value(network_params, ck.values)
    return ck.updater_values, ck.epoch


def find_reload_model(args):
    """Resume from <save_path>_last_model.pkl unless a model was named or --no-reload given."""
    if args.reload_model or args.no_reload:
        return
    candidate = args.save_path + '_last_model.pkl'
    if os.path.exists(candidate):
        print('Previously trained model detected: {}'.format(candidate))
        print('Training continues')
        args.reload_model = candidate


# ------------------------------------------------------------------------------------------------
# evaluation history
class EvalHistory(list):
    """List of EvalRecord; starts with the reference's sentinel row (scripts/exp_deep_lstm2.py:113)."""
    SENTINEL = EvalRecord(10.1, 10.0, 1.0, 10.0, 1.0)

    def __init__(self, records=None):
        list.__init__(self, [self.SENTINEL] if records is None else records)

    def best_valid_fer(self):
        return min(r.valid_fer for r in self)

    def add(self, train_ce, valid, test):
        """valid / test: (ce_frame, fer).  Returns True when this epoch is the best so far."""
        improved = valid[1] < self.best_valid_fer()
        self.append(EvalRecord(train_ce, valid[0], valid[1], test[0], test[1]))
        return improved

    def save(self, path):
        with open(path, 'wb') as f:
            pickle.dump([tuple(r) for r in self], f, protocol=2)

    @classmethod
    def load(cls, path):
        return cls([EvalRecord(*r) for r in _read_data_pickle(path)])


def save_eval_history(eval_history, save_path):
    EvalHistory(list(eval_history)).save(save_path)


def best_fer(eval_history):
    return min(r.valid_fer for r in eval_history)


# ------------------------------------------------------------------------------------------------
# status lines
def show_status(save_path, ce_frame, network_grads_norm, batch_idx, batch_size, epoch_idx):
    print('--')
    print('Model Name: {} (Epoch {})'.format(os.path.basename(save_path), epoch_idx))
    print('Train CE {} (batch {}, {} examples so far): '.format(ce_frame, batch_idx, batch_idx * batch_size))
    print('Gradient Norm: {}'.format(network_grads_norm))


class StopWatch(object):
    def __init__(self):
        self.reset()

    def reset(self):
        self.t0 = time.time()

    def elapsed(self):
        return time.time() - self.t0

    def print_elapsed(self):
        s = self.elapsed()
        print('Took {:.1f}s ({:.1f}m; {:.1f}h)'.format(s, s / 60, s / 3600))


# ------------------------------------------------------------------------------------------------
# TBPTT windows
def windows(n_seq, win_size, right_context=0):
    """Slices [i, i + win_size + right_context) for i = 0, win_size, ...: the look-ahead frames are
    processed again by the next window (numpy clips the last slice)."""
    return [slice(i, i + win_size + right_context) for i in range(0, n_seq, win_size)]


def _cut(batch, n_items):
    arrays = tuple(batch)
    if len(arrays) != n_items:
        raise ValueError('expected a %d-tuple batch, got %d items' % (n_items, len(arrays)))
    return arrays, arrays[0].shape[1]


def gen_win(batch, win_size, right_context=0):
    """batch = (x, x_mask, ivec, ivec_mask, y, y_mask); absent i-vector entries stay None."""
    arrays, n_seq = _cut(batch, 6)
    for s in windows(n_seq, win_size, right_context):
        yield tuple(None if a is None else a[:, s] for a in arrays)


def gen_win_test(batch, win_size, right_context=0):
    """The 4-tuple form of the forward scripts: (x, x_mask, ivec, ivec_mask)."""
    arrays, n_seq = _cut(batch, 4)
    for s in windows(n_seq, win_size, right_context):
        yield tuple(None if a is None else a[:, s] for a in arrays)

"""Optimiser factories with the call shape of libs/lasagne_libs/updates.py.  The reference's updaters
take symbolic gradients and return (updates OrderedDict, opt_params_dict); here the "updates" object
is an `Update` whose `.apply(network)` runs the fused HIP optimiser kernel on the network's flat
parameter / gradient buffers, and opt_params_dict exposes the same keys ('t_prev', and for Adam
'm_prev' / 'v_prev').

Only `adam` (libs/lasagne_libs/updates.py:534-618: what every hot-path script passes,
scripts/exp_deep_lstm2.py:17,85) is on the path; the other optimisers are SURVEY 8f rank 4.
"""
from collections import OrderedDict

import numpy as np


class SharedScalar(object):
    """Minimal stand-in for `theano.shared(floatX(x))`: .get_value() / .set_value()."""
    def __init__(self, value, name=None):
        self.value = np.float32(value); self.name = name

    def get_value(self):
        return self.value

    def set_value(self, v):
        self.value = np.float32(v)


class Update(object):
    def apply(self, network, want_sqrt=True):
        raise NotImplementedError


class AdamUpdate(Update):
    def __init__(self, learning_rate, beta1, beta2, epsilon, t_prev):
        self.learning_rate = learning_rate
        self.beta1, self.beta2, self.epsilon = beta1, beta2, epsilon
        self.t_prev = t_prev

    def _lr(self):
        lr = self.learning_rate
        return float(lr.get_value()) if hasattr(lr, 'get_value') else float(lr)

    def apply(self, network, want_sqrt=True):
        network.t_prev = int(round(float(self.t_prev.get_value())))
        norm = network.adam_update(self._lr(), self.beta1, self.beta2, self.epsilon, want_sqrt=want_sqrt)
        self.t_prev.set_value(network.t_prev)
        return norm


def adam(loss_or_grads=None, params=None, learning_rate=0.001, beta1=0.9, beta2=0.999, epsilon=1e-8,
         load_params_dict=None):
    opt_params_dict = OrderedDict()
    t0 = 0.0
    if load_params_dict:
        t0 = float(np.asarray(load_params_dict['t_prev']))
    t_prev = SharedScalar(t0, 't_prev')
    opt_params_dict['t_prev'] = t_prev
    return AdamUpdate(learning_rate, beta1, beta2, epsilon, t_prev), opt_params_dict

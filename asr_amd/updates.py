"""Optimiser factories with the call shape of libs/lasagne_libs/updates.py.  The reference's updaters
take symbolic gradients and return (updates OrderedDict, opt_params_dict); here the "updates" object
is an `Update` whose `.apply(network)` runs the fused HIP optimiser kernel on the network's flat
parameter / gradient buffers, and opt_params_dict exposes the same keys ('t_prev', and for Adam
'm_prev' / 'v_prev').

`adam` (libs/lasagne_libs/updates.py:534-618) is what every hot-path script passes
(scripts/exp_deep_lstm2.py:17,85); the scripts around models/gating_hyper_nets.py pick
`eval(args.updater)` among momentum / adam / adamax / nesterov_momentum and clip with
`total_norm_constraint` (scripts/dln/ted/exp_prj_lstm_dln.py:17,135-148).  All of them run as the one
elementwise HIP kernel asr_opt_step; the state tensors are flat, like the parameters.
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
    if isinstance(loss_or_grads, ClippedGrads):        # adam behind total_norm_constraint
        return OptUpdate(7, learning_rate, beta1, beta2, epsilon, loss_or_grads, t_prev, beta1), opt_params_dict
    return AdamUpdate(learning_rate, beta1, beta2, epsilon, t_prev), opt_params_dict


class ClippedGrads(object):
    """What total_norm_constraint(..., return_norm=True)[0] stands for here: "the network's gradients,
    scaled so that their global norm is at most max_norm" -- applied inside the optimiser kernel."""
    def __init__(self, max_norm, epsilon):
        self.max_norm = float(max_norm); self.epsilon = float(epsilon)


class NormHandle(object):
    """Second return value of total_norm_constraint(return_norm=True): filled by the next update."""
    value = None


def total_norm_constraint(tensor_vars=None, max_norm=0.0, epsilon=1e-7, return_norm=False):
    """libs/lasagne_libs/updates.py:781-855 (lasagne.updates.total_norm_constraint)."""
    cg = ClippedGrads(max_norm, epsilon)
    return (cg, NormHandle()) if return_norm else cg


class OptUpdate(Update):
    """sgd / momentum / nesterov / adagrad / rmsprop / adadelta / adamax on asr_opt_step."""
    def __init__(self, kind, learning_rate, h1, h2, eps, grads, t_prev=None, beta1=None):
        self.kind = kind; self.learning_rate = learning_rate
        self.h1, self.h2, self.eps = h1, h2, eps
        self.t_prev = t_prev; self.beta1 = beta1
        self.clip = grads if isinstance(grads, ClippedGrads) else None

    def _lr(self):
        lr = self.learning_rate
        return float(lr.get_value()) if hasattr(lr, 'get_value') else float(lr)

    def apply(self, network, want_sqrt=True):
        lr = self._lr()
        if self.kind == 6:                       # adamax: a_t = lr / (1 - beta1^t), :662-664
            t = int(round(float(self.t_prev.get_value()))) + 1
            self.t_prev.set_value(t)
            lr = lr / (1.0 - self.beta1 ** t)
        elif self.kind == 7:                     # adam with gradient clipping
            t = int(round(float(self.t_prev.get_value()))) + 1
            self.t_prev.set_value(t)
            network.t_prev = t
            lr = lr * np.sqrt(1.0 - self.h2 ** t) / (1.0 - self.h1 ** t)
        mg = self.clip.max_norm if self.clip else 0.0
        ne = self.clip.epsilon if self.clip else 1e-7
        return network.opt_update(self.kind, lr, self.h1, self.h2, self.eps, mg, ne, want_sqrt=want_sqrt)


def _mk(kind, grads, learning_rate, h1=0.0, h2=0.0, eps=0.0, with_t=False, load_params_dict=None):
    d = OrderedDict()
    t_prev = None
    if with_t:
        t_prev = SharedScalar(float(np.asarray(load_params_dict['t_prev'])) if load_params_dict else 0.0, 't_prev')
        d['t_prev'] = t_prev
    return OptUpdate(kind, learning_rate, h1, h2, eps, grads, t_prev, h1 if with_t else None), d


def sgd(loss_or_grads=None, params=None, learning_rate=0.0, load_params_dict=None):
    """:64-92."""
    return _mk(0, loss_or_grads, learning_rate)


def momentum(loss_or_grads=None, params=None, learning_rate=0.0, momentum=0.9, load_params_dict=None):
    """:152-188 (sgd + apply_momentum :95-149)."""
    return _mk(1, loss_or_grads, learning_rate, momentum)


def nesterov_momentum(loss_or_grads=None, params=None, learning_rate=0.0, momentum=0.9, load_params_dict=None):
    """:254-295 (sgd + apply_nesterov_momentum :191-251)."""
    return _mk(2, loss_or_grads, learning_rate, momentum)


def adagrad(loss_or_grads=None, params=None, learning_rate=1.0, epsilon=1e-6, load_params_dict=None):
    """:298-363."""
    return _mk(3, loss_or_grads, learning_rate, eps=epsilon)


def rmsprop(loss_or_grads=None, params=None, learning_rate=1.0, rho=0.9, epsilon=1e-6, load_params_dict=None):
    """:366-433."""
    return _mk(4, loss_or_grads, learning_rate, rho, eps=epsilon)


def adadelta(loss_or_grads=None, params=None, learning_rate=1.0, rho=0.95, epsilon=1e-6, load_params_dict=None):
    """:436-531."""
    return _mk(5, loss_or_grads, learning_rate, rho, eps=epsilon)


def adamax(loss_or_grads=None, params=None, learning_rate=0.002, beta1=0.9, beta2=0.999, epsilon=1e-8,
           load_params_dict=None):
    """:621-698."""
    return _mk(6, loss_or_grads, learning_rate, beta1, beta2, epsilon, with_t=True, load_params_dict=load_params_dict)

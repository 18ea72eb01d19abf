"""Shared helpers for the tests: seeded parameter / batch generation."""
import numpy as np


def init_value(rng, shape, init, perturb=0.05):
    """Reference init specs (asr_amd/specs.py) followed by + N(0, perturb) on every constant
    -initialised tensor, so that the adaptive paths (hyper d == 1, LHUC scale == 1, cond-LN
    alpha == 1) are not exact identities."""
    if init == 'orthogonal':
        a = rng.standard_normal(shape)
        if len(shape) == 2:
            u, _, vt = np.linalg.svd(a, full_matrices=False)
            a = u if u.shape == tuple(shape) else vt
        return a
    if init == 'glorot_uniform':
        lim = np.sqrt(6.0 / (shape[0] + shape[1]))
        return rng.uniform(-lim, lim, shape)
    kind, v = init
    if kind == 'normal':
        return rng.standard_normal(shape) * v
    if kind == 'uniform':
        return rng.uniform(-v, v, shape) + rng.standard_normal(shape) * perturb
    return np.full(shape, v, dtype=np.float64) + rng.standard_normal(shape) * perturb


def make_params(rng, spec, perturb=0.05, scale=1.0):
    return {n: init_value(rng, s, i, perturb) * (scale if i == 'orthogonal' else 1.0)
            for n, s, i in spec}


def make_batch(rng, B, T, D, C, lengths=None, I=0):
    if lengths is None:
        lengths = [T] + [1] + list(rng.integers(2, T + 1, size=B - 2)) if B >= 2 else [T]
    lengths = np.asarray(lengths[:B])
    x = rng.standard_normal((B, T, D)).astype(np.float32)
    mask = (np.arange(T)[None, :] < lengths[:, None]).astype(np.float32)
    x *= mask[:, :, None]
    y = rng.integers(0, C, size=(B, T)).astype(np.int32)
    out = dict(x=x, mask=mask, y=y, lengths=lengths)
    if I:
        iv = rng.standard_normal((B, 1, I)).astype(np.float32)
        out['ivec'] = np.repeat(iv, T, axis=1)
    return out

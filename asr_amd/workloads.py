"""Seeded synthetic batches shaped like the five BASELINE configs: features
i.i.d. N(0,1) float32, targets uniform int32, masks from lengths (left-aligned, zero padded as
Fuel's Padding does), i-vectors constant within an utterance."""
import numpy as np


def _lognormal_lengths(rng, n, mu, sigma, lo, hi):
    return np.clip(np.round(rng.lognormal(np.log(mu), sigma, size=n)), lo, hi).astype(np.int64)


def lengths_for(cfg, rng, B):
    if cfg == 1:
        return rng.integers(90, 781, size=B)
    if cfg == 2:
        return _lognormal_lengths(rng, B, 760, 0.35, 150, 2000)
    if cfg == 3:      # two concatenated utterances (README.md:6-7)
        return _lognormal_lengths(rng, B, 760, 0.35, 150, 2000) + _lognormal_lengths(rng, B, 760, 0.35, 150, 2000)
    if cfg == 4:
        return _lognormal_lengths(rng, B, 800, 0.5, 100, 2500)
    if cfg == 5:
        return rng.integers(2000, 6001, size=B)
    raise KeyError(cfg)


CFG = {1: dict(D=123, C=61, seed=1001, I=0), 2: dict(D=123, C=3436, seed=1002, I=0),
       3: dict(D=123, C=3436, seed=1003, I=0), 4: dict(D=123, C=4174, seed=1004, I=100),
       5: dict(D=123, C=3436, seed=1005, I=100)}


def length_sorted_batches(cfg, B, n_batches, cache=1024, seed_offset=0):
    """Utterance lengths of `n_batches` batches of B rows drawn the way the reference's LengthSortTransformer forms
    them (data/transformers.py:15-63; create_ivector_datastream(..., length_sort=True, min_after_cache=1024),
    data/fuel_utils.py:94-109): a cache of >= `cache` shuffled utterances is sorted by length and cut into consecutive
    batches.  The batches of a cache are then visited in a seeded random order, so that a run does not see them
    shortest first.  Returns a list of int64 arrays."""
    c = CFG[cfg]
    rng = np.random.default_rng(c['seed'] + 7919 + seed_offset)
    out = []
    while len(out) < n_batches:
        n = max(cache, B)
        n = ((n + B - 1) // B) * B
        lens = np.sort(lengths_for(cfg, rng, n), kind='stable')
        cuts = [lens[i:i + B] for i in range(0, n, B)]
        rng.shuffle(cuts)
        out.extend(cuts)
    return out[:n_batches]


def make_batch(cfg, B=16, seed_offset=0, max_len=None, lengths=None):
    """lengths: take these utterance lengths (length_sorted_batches, or a rank's share of a global batch) instead of
    drawing B fresh ones; features / targets are seeded by seed_offset as before."""
    c = CFG[cfg]
    rng = np.random.default_rng(c['seed'] + seed_offset)
    lens = lengths_for(cfg, rng, B) if lengths is None else np.asarray(lengths, np.int64)
    B = len(lens)
    if max_len:
        lens = np.minimum(lens, max_len)
    T = int(lens.max())
    x = rng.standard_normal((B, T, c['D'])).astype(np.float32)
    mask = (np.arange(T)[None, :] < lens[:, None]).astype(np.float32)
    x *= mask[:, :, None]
    y = rng.integers(0, c['C'], size=(B, T)).astype(np.int32)
    out = dict(x=x, mask=mask, y=y, lengths=lens, frames=int(lens.sum()))
    if c['I']:
        iv = rng.standard_normal((B, 1, c['I'])).astype(np.float32)
        out['ivec'] = np.ascontiguousarray(np.repeat(iv, T, axis=1))
    return out

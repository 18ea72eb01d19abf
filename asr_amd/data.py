"""Batch streams with the tuple layout the reference's Fuel streams yield
(libs/comp_graph_utils.py:179; data/fuel_utils.py:94-123):
    (input_data f32 (B,T,D), input_mask f32 (B,T), ivector_data f32 (B,T,I), ivector_mask f32 (B,T),
     target_data i32 (B,T), target_mask f32 (B,T))
left-aligned, zero padded to the longest utterance of the batch as Fuel's `Padding` transformer does.
The reference reads Fuel HDF5 files (h5py / Fuel are absent here and outside the hot path); these
streams are fed from Kaldi archives (asr_amd.kaldi_io) or from in-memory utterance lists.
"""
import numpy as np

from . import kaldi_io


class Utterance(object):
    __slots__ = ('uttid', 'features', 'ivectors', 'targets')

    def __init__(self, uttid, features, ivectors=None, targets=None):
        self.uttid = uttid
        self.features = np.asarray(features, dtype=np.float32)
        T = self.features.shape[0]
        if ivectors is not None:
            iv = np.asarray(ivectors, dtype=np.float32)
            if iv.ndim == 1:                      # one i-vector per utterance: repeat over frames
                iv = np.repeat(iv[None, :], T, axis=0)
            ivectors = iv
        self.ivectors = ivectors
        self.targets = None if targets is None else np.asarray(targets, dtype=np.int32)


def _pad(seqs, dtype, tail_shape=()):
    B = len(seqs)
    T = max(s.shape[0] for s in seqs)
    out = np.zeros((B, T) + tuple(tail_shape), dtype=dtype)
    mask = np.zeros((B, T), dtype=np.float32)
    for i, s in enumerate(seqs):
        out[i, :s.shape[0]] = s
        mask[i, :s.shape[0]] = 1.0
    return out, mask


class UtteranceStream(object):
    """get_epoch_iterator() yields 6-tuples (train/eval) or 4-tuples (`test=True`, as
    create_ivector_test_datastream does, bin/forward_lstm2.py:78)."""

    def __init__(self, utterances, batch_size, test=False, sort_by_length=False):
        self.utts = list(utterances)
        if sort_by_length:                        # LengthSortTransformer, data/transformers.py
            self.utts.sort(key=lambda u: u.features.shape[0])
        self.batch_size = batch_size
        self.test = test

    def __len__(self):
        return (len(self.utts) + self.batch_size - 1) // self.batch_size

    def get_epoch_iterator(self):
        for i in range(0, len(self.utts), self.batch_size):
            chunk = self.utts[i:i + self.batch_size]
            x, xm = _pad([u.features for u in chunk], np.float32, chunk[0].features.shape[1:])
            if chunk[0].ivectors is not None:
                iv, ivm = _pad([u.ivectors for u in chunk], np.float32, chunk[0].ivectors.shape[1:])
            else:
                iv, ivm = None, None
            if self.test:
                yield (x, xm, iv, ivm)
                continue
            y, ym = _pad([u.targets for u in chunk], np.int32)
            yield (x, xm, iv, ivm, y, ym)


class UttIdStream(object):
    """get_uttid_stream (data/wsj/fuel_utils.py): yields 1-tuples of uttid lists, batch-aligned."""

    def __init__(self, utterances, batch_size):
        self.ids = [u.uttid for u in utterances]
        self.batch_size = batch_size

    def get_epoch_iterator(self):
        for i in range(0, len(self.ids), self.batch_size):
            yield (self.ids[i:i + self.batch_size],)


def read_int_vectors(path):
    """Kaldi text int-vector archive: `uttid 3 1 4 ...` per line (ali-to-pdf | copy-int-vector ark,t)."""
    out = {}
    with open(path) as f:
        for line in f:
            parts = line.split()
            if parts:
                out[parts[0]] = np.asarray([int(v) for v in parts[1:]], dtype=np.int32)
    return out


def load_kaldi_set(feat_rspecifier, targets_path=None, ivector_rspecifier=None):
    ivecs = dict(kaldi_io.read_mat_ark(ivector_rspecifier)) if ivector_rspecifier else {}
    tgts = read_int_vectors(targets_path) if targets_path else {}
    utts = []
    for key, mat in kaldi_io.SequentialBaseFloatMatrixReader(feat_rspecifier):
        iv = ivecs.get(key)
        if iv is not None and iv.shape[0] == 1:
            iv = iv[0]
        utts.append(Utterance(key, mat, iv, tgts.get(key)))
    return utts


def synthetic_set(n_utts, D, C, I=0, min_len=20, max_len=60, seed=0):
    rng = np.random.default_rng(seed)
    utts = []
    for k in range(n_utts):
        T = int(rng.integers(min_len, max_len + 1))
        utts.append(Utterance('utt%04d' % k, rng.standard_normal((T, D)).astype(np.float32),
                              rng.standard_normal(I).astype(np.float32) if I else None,
                              rng.integers(0, C, size=T).astype(np.int32)))
    return utts

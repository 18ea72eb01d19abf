"""Data-side transformers that change what the hot path sees, on plain Python
iterators of example lists instead of Fuel streams:

  DelayTransformer        data/transformers.py:133-177   features / ivectors repeat their last frame
                          `delay` times at the END, targets get `delay` zeros at the FRONT
  LengthSortTransformer   data/transformers.py:15-63     cache >= min_after_cache examples, sort the
                          cache by length, serve batches of batch_size from its front
  skip_frames             libs/utils.py:208-229          every n-th frame (or a random one per group)
  skip_frames_fixed       libs/utils.py:231-262          frames start::n, start 0 or random in [0,n)
  unskip                  bin/forward_lstm2.py:99-101    numpy.repeat(net_output, n, axis=1)[:, :n_seq]

A "batch" here is what the wrapped Fuel stream hands those classes: a tuple of sources, each a list of
per-utterance arrays (not yet padded)."""
import random

import numpy as np


class DelayTransformer(object):
    """sources order as the reference returns it: (features, ivectors, targets)."""
    def __init__(self, data_stream, delay, sources=('features', 'ivectors', 'targets')):
        self.data_stream = data_stream
        self.delay = int(delay)
        self.sources = tuple(sources)

    def _delayed_feat(self, feat_batch):
        out = []
        for ex in feat_batch:
            ex = np.asarray(ex)
            out.append(np.concatenate([ex, np.tile(ex[-1, :], (self.delay, 1))]))
        return out

    def _delayed_targets(self, target_batch):
        return [np.concatenate([np.zeros((self.delay,), dtype=np.int32), np.asarray(ex)]) for ex in target_batch]

    def transform_batch(self, batch):
        idx = {s: i for i, s in enumerate(self.sources)}
        return [self._delayed_feat(batch[idx['features']]), self._delayed_feat(batch[idx['ivectors']]),
                self._delayed_targets(batch[idx['targets']])]

    def get_epoch_iterator(self):
        for batch in self.data_stream.get_epoch_iterator():
            yield self.transform_batch(batch)


class LengthSortTransformer(object):
    def __init__(self, data_stream, batch_size, min_after_cache):
        if min_after_cache < batch_size:
            raise ValueError('capacity is smaller than batch size')
        self.data_stream = data_stream
        self.batch_size = batch_size
        self.min_after_cache = min_after_cache

    def get_epoch_iterator(self):
        child = iter(self.data_stream.get_epoch_iterator())
        cache = None
        exhausted = False
        while True:
            if cache is None or self.batch_size > len(cache[0]):
                # _cache(): refill up to min_after_cache, then sort EACH source by its own lengths
                # (:58-63 sorts the sources independently, as the reference does)
                while not exhausted and (cache is None or len(cache[0]) < self.min_after_cache):
                    try:
                        data = next(child)
                    except StopIteration:
                        exhausted = True
                        break
                    if cache is None:
                        cache = [[] for _ in data]
                    for c, d in zip(cache, data):
                        c.extend(d)
                if cache is None or not cache[0]:
                    return
                for c in cache:
                    c.sort(key=lambda x: len(x))
            yield tuple(c[:self.batch_size] for c in cache)
            cache = [c[self.batch_size:] for c in cache]
            if exhausted and not cache[0]:
                return


def skip_frames(batch, every_n, random_choice=False):
    new_batch = []
    for src_data in batch:
        new_src = []
        for ex in src_data:
            ex = np.asarray(ex)
            n_seq = ex.shape[0]
            if random_choice:
                idx = [min(i + random.randint(0, every_n - 1), n_seq - 1) for i in range(0, n_seq, every_n)]
                new_src.append(ex[idx])
            else:
                new_src.append(ex[::every_n])
        new_batch.append(new_src)
    return new_batch


def skip_frames_fixed(batch, every_n, return_first=False):
    assert every_n > 0
    if every_n == 1:
        yield batch
        return
    starts = range(1) if return_first else [random.randint(0, every_n - 1)]
    for start in starts:
        new_batch = []
        for src_data in batch:
            new_src = []
            for ex in src_data:
                ex = np.asarray(ex)
                assert not ex.shape[0] < every_n
                new_src.append(ex[start::every_n])
            new_batch.append(new_src)
        yield new_batch


def unskip(net_output, every_n, n_seq):
    return np.repeat(net_output, every_n, axis=1)[:, :n_seq]

"""Kaldi float-matrix archives: the posterior-file contract of bin/forward_lstm2.py:74,105-107 and
bin/forward_ln_lstmp.py:103,116-118, which write one (valid_len, C) matrix of log-posteriors per
utterance through `kaldi_io.BaseFloatMatrixWriter(wxfilename)` (third-party package, absent here) into a
Kaldi wspecifier that decode_wsj.sh:24 pipes into `latgen-faster-mapped`.

Format written (Kaldi's binary table format, BaseFloat = float32):
    <key> ' ' '\\0' 'B' 'F' 'M' ' ' '\\4' <int32 rows> '\\4' <int32 cols> <rows*cols float32, row-major, LE>
Text mode (",t"):  <key> '  [' newline, one row per line, last row closed by ' ]'.

Supported specifiers:  ark:PATH | ark:- | ark,t:PATH | ark,scp:ARK,SCP | scp:PATH (read) .
Readers also accept double matrices ('DM') and return float32.
"""
import struct
import sys

import numpy as np


def _parse_spec(spec):
    if ':' not in spec:
        raise ValueError('not a Kaldi table specifier: %r' % (spec,))
    kinds, rest = spec.split(':', 1)
    opts = [k.strip() for k in kinds.split(',')]
    return opts, rest


class BaseFloatMatrixWriter(object):
    """`writer = BaseFloatMatrixWriter('ark:-'); writer.write(uttid, matrix); writer.close()`."""

    def __init__(self, wspecifier):
        opts, rest = _parse_spec(wspecifier)
        self.text = 't' in opts
        self.scp = None
        self._own = False
        if 'ark' not in opts:
            raise ValueError('wspecifier must contain "ark": %r' % (wspecifier,))
        if 'scp' in opts:
            ark_path, scp_path = rest.split(',', 1)
            self.scp = open(scp_path, 'w')
        else:
            ark_path = rest
        self.ark_path = ark_path
        if ark_path == '-':
            self.f = getattr(sys.stdout, 'buffer', sys.stdout)
        else:
            self.f = open(ark_path, 'wb')
            self._own = True
        self.closed = False

    def write(self, key, value):
        if isinstance(key, bytes):
            key = key.decode('ascii')
        if not key or any(c.isspace() for c in key):
            raise ValueError('invalid Kaldi key %r' % (key,))
        m = np.ascontiguousarray(value, dtype='<f4')
        if m.ndim != 2:
            raise ValueError('BaseFloatMatrixWriter.write expects a 2-D matrix, got shape %r' % (m.shape,))
        f = self.f
        f.write((key + ' ').encode('ascii'))
        if self.text:
            f.write(b' [')
            if m.shape[0] == 0:
                f.write(b' ]\n')
            for i, row in enumerate(m):
                f.write(b'\n  ' + ' '.join(repr(float(v)) for v in row).encode('ascii'))
                if i == m.shape[0] - 1:
                    f.write(b' ]\n')
            return
        if self.scp is not None:
            self.scp.write('%s %s:%d\n' % (key, self.ark_path, f.tell()))
        f.write(b'\0B')
        f.write(b'FM ')
        f.write(b'\4' + struct.pack('<i', m.shape[0]))
        f.write(b'\4' + struct.pack('<i', m.shape[1]))
        f.write(m.tobytes())

    def flush(self):
        self.f.flush()
        if self.scp is not None:
            self.scp.flush()

    def close(self):
        if self.closed:
            return
        self.flush()
        if self._own:
            self.f.close()
        if self.scp is not None:
            self.scp.close()
        self.closed = True

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()


def _read_token(f):
    out = b''
    while True:
        c = f.read(1)
        if not c:
            return None if not out else out.decode('ascii')
        if c in b' \t\n':
            if out:
                return out.decode('ascii')
            continue
        out += c


def _read_binary_matrix(f):
    tok = f.read(3)
    if tok not in (b'FM ', b'DM '):
        raise ValueError('unsupported Kaldi matrix type %r' % (tok,))
    dt = '<f4' if tok == b'FM ' else '<f8'
    dims = []
    for _ in range(2):
        if f.read(1) != b'\4':
            raise ValueError('bad Kaldi int header')
        dims.append(struct.unpack('<i', f.read(4))[0])
    n = dims[0] * dims[1]
    buf = f.read(n * np.dtype(dt).itemsize)
    if len(buf) != n * np.dtype(dt).itemsize:
        raise ValueError('truncated Kaldi matrix')
    return np.frombuffer(buf, dtype=dt).reshape(dims).astype(np.float32)


def _read_text_matrix(f):
    rows = []
    cur = b''
    # '[' has been consumed by the caller
    while True:
        c = f.read(1)
        if not c:
            raise ValueError('unterminated Kaldi text matrix')
        if c == b']':
            if cur.strip():
                rows.append([float(v) for v in cur.split()])
            f.readline()
            break
        if c == b'\n':
            if cur.strip():
                rows.append([float(v) for v in cur.split()])
            cur = b''
        else:
            cur += c
    if not rows:
        return np.zeros((0, 0), dtype=np.float32)
    return np.asarray(rows, dtype=np.float32)


def _read_matrix(f):
    c = f.read(1)
    while c in (b' ', b'\t'):
        c = f.read(1)
    if c == b'\0':
        if f.read(1) != b'B':
            raise ValueError('bad Kaldi binary marker')
        return _read_binary_matrix(f)
    if c == b'[':
        return _read_text_matrix(f)
    raise ValueError('unrecognised Kaldi matrix start %r' % (c,))


class SequentialBaseFloatMatrixReader(object):
    """Iterates (key, float32 matrix) over `ark:PATH`, `ark:-` or `scp:PATH`."""

    def __init__(self, rspecifier):
        opts, rest = _parse_spec(rspecifier)
        self.opts, self.rest = opts, rest

    def __iter__(self):
        if 'scp' in self.opts and 'ark' not in self.opts:
            with open(self.rest) as scp:
                for line in scp:
                    line = line.strip()
                    if not line:
                        continue
                    key, loc = line.split(None, 1)
                    path, _, off = loc.rpartition(':')
                    if not path:
                        path, off = loc, '0'
                    with open(path, 'rb') as f:
                        f.seek(int(off))
                        yield key, _read_matrix(f)
            return
        f = getattr(sys.stdin, 'buffer', sys.stdin) if self.rest == '-' else open(self.rest, 'rb')
        try:
            while True:
                key = _read_token(f)
                if key is None:
                    break
                yield key, _read_matrix(f)
        finally:
            if self.rest != '-':
                f.close()


def read_mat_ark(rspecifier):
    return list(SequentialBaseFloatMatrixReader(rspecifier))

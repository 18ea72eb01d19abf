"""Posterior forward passes: the five bin/forward_*.py entry points that matter.

  forward_lstm2          bin/forward_lstm2.py:25-109 (and bin/forward_lstm.py, whose builder import is
                         broken in the reference): build_deep_lstm -> ff -> log(posteriors)
  forward_lstm2_tbptt    bin/forward_lstm2_tbptt.py: windowed forward with carried state
  forward_ln_lstmp       bin/forward_ln_lstmp.py:55-118: logits -> in-graph log-softmax
  forward_ivec_ln_lstmp  bin/forward_ivec_ln_lstmp.py:58-125: (x, cond, mask) argument order

Each writes, per utterance, a (valid_len, C) natural-log posterior matrix keyed by uttid into the Kaldi
wspecifier `wxfilename` (e.g. `ark:-` piped into latgen-faster-mapped, decode_wsj.sh:24).

CLI:  python -m asr_amd.forward {lstm2|lstm2_tbptt|ln_lstmp|ivec_ln_lstmp} [flags] MODEL FEATS WXFILENAME
      FEATS is a Kaldi feature rspecifier (the reference takes a Fuel HDF5 dataset name; data formats
      are SURVEY 8f rank 2); --ivectors gives the i-vector rspecifier.
"""
from __future__ import print_function

import argparse
import sys

import numpy as np

from . import kaldi_io
from .arg_utils import get_arg_parser
from .builders import build_deep_lstm, build_deep_lstm_tbptt
from .comp_graph_utils import ff, ff_cond, ff_tbptt
from .data import UtteranceStream, UttIdStream, load_kaldi_set
from .gating_hyper_nets import deep_projection_ln_lstm_model, deep_projection_ivector_ln_model
from .utils import gen_win_test, load_network_pickle, set_model_param_value


def load_model(network, path):
    vals, _, _ = load_network_pickle(path)
    set_model_param_value(network.get_all_params(trainable=True), vals)


def write_posteriors(writer, net_output, input_mask, uttids, take_log):
    feat_lens = input_mask.sum(axis=1)
    for out_idx, (output, uttid) in enumerate(zip(net_output, uttids)):
        valid_len = int(feat_lens[out_idx])
        m = output[:valid_len]
        with np.errstate(divide='ignore'):
            writer.write(uttid, np.log(m) if take_log else m)


def skip_frames_test(feat_batch, every_n):
    """libs/utils.py:208-229, deterministic variant: keep every n-th frame."""
    return tuple(None if a is None else a[:, ::every_n] for a in feat_batch)


def run_forward_lstm2(network, args, test_ds, uttid_stream, writer):
    ff_fn = ff(network, True, True, True if args.use_ivector_input else None)
    for feat_batch, (uttid_batch,) in zip(test_ds.get_epoch_iterator(), uttid_stream.get_epoch_iterator()):
        input_data, input_mask, ivector_data, ivector_mask = feat_batch
        n_seq = input_data.shape[1]
        if getattr(args, 'skip', 0):
            s_x, s_m, s_iv, _ = skip_frames_test(feat_batch, args.skip)
            net_output, = ff_fn(s_x, s_m, s_iv) if args.use_ivector_input else ff_fn(s_x, s_m)
            net_output = np.repeat(net_output, args.skip, axis=1)[:, :n_seq, :]
        elif args.use_ivector_input:
            net_output, = ff_fn(input_data, input_mask, ivector_data)
        else:
            net_output, = ff_fn(input_data, input_mask)
        write_posteriors(writer, net_output, input_mask, uttid_batch, take_log=True)


def run_forward_lstm2_tbptt(network, tbptt_layers, args, test_ds, uttid_stream, writer):
    """The reference's ff_tbptt compiles no state updates, so its windows all start from zero state
    (libs/comp_graph_utils.py:335-351); we follow the layer semantics (state carried between the
    windows of a batch, libs/lhuclstm_layers.py:313-319) as predictor_tbptt does."""
    ff_fn = ff_tbptt(True, True, network, True, args.delay, args.num_tbptt_steps,
                     True if args.use_ivector_input else None)
    for feat_batch, (uttid_batch,) in zip(test_ds.get_epoch_iterator(), uttid_stream.get_epoch_iterator()):
        i_data, i_mask = feat_batch[0], feat_batch[1]
        for l in tbptt_layers:
            l.reset(i_data.shape[0])
        outs = []
        for win_idx, win in enumerate(gen_win_test(feat_batch, args.num_tbptt_steps, args.right_context), 1):
            x, m, iv, _ = win
            first = 1 if win_idx == 1 else 0
            o, = ff_fn(x, m, iv, first) if args.use_ivector_input else ff_fn(x, m, first)
            outs.append(o)
        net_output = np.concatenate(outs, axis=1)
        write_posteriors(writer, net_output, i_mask, uttid_batch, take_log=True)


def run_forward_ln_lstmp(network, args, test_ds, uttid_stream, writer):
    ff_fn = ff(network, True, True, None, log=True)
    for feat_batch, (uttid_batch,) in zip(test_ds.get_epoch_iterator(), uttid_stream.get_epoch_iterator()):
        input_data, input_mask = feat_batch[0], feat_batch[1]
        net_output, = ff_fn(input_data, input_mask)
        write_posteriors(writer, net_output, input_mask, uttid_batch, take_log=False)


def run_forward_ivec_ln_lstmp(network, args, test_ds, uttid_stream, writer):
    ff_fn = ff_cond(network, True, True, True, log=True)
    for feat_batch, (uttid_batch,) in zip(test_ds.get_epoch_iterator(), uttid_stream.get_epoch_iterator()):
        input_data, input_mask, input_cond, _ = feat_batch
        net_output, = ff_fn(input_data, input_cond, input_mask)
        write_posteriors(writer, net_output, input_mask, uttid_batch, take_log=False)


def _hyper_args():
    """bin/forward_ln_lstmp.py:20-46 / bin/forward_ivec_ln_lstmp.py:20-49 (underscore flags)."""
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    for name, typ, default in (('--batch_size', int, 16), ('--num_layers', int, 3), ('--num_units', int, 512),
                               ('--num_factors', int, 64), ('--num_conds', int, 3), ('--grad_clipping', float, 1.0),
                               ('--dropout', float, 0.0), ('--input_dim', int, 123), ('--output_dim', int, 3436),
                               ('--ivector_dim', int, 100)):
        p.add_argument(name, default=default, type=typ)
    return p


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    if not argv:
        print(__doc__, file=sys.stderr)
        return 2
    which = argv.pop(0)
    if which in ('lstm2', 'lstm', 'lstm2_tbptt'):
        parser = get_arg_parser('deep_lstm')
    elif which in ('ln_lstmp', 'ivec_ln_lstmp'):
        parser = _hyper_args()
    else:
        print('unknown forward script %r' % which, file=sys.stderr)
        return 2
    parser.add_argument('--ivectors', default=None, help='Kaldi rspecifier of per-utterance i-vectors')
    parser.add_argument('--device', default='cuda:0')
    parser.add_argument('model')
    parser.add_argument('dataset', help='Kaldi feature rspecifier, e.g. ark:feats.ark')
    parser.add_argument('wxfilename')
    args = parser.parse_args(argv)
    print(args, file=sys.stderr)
    if not args.model:
        print('Must specfiy network to load', file=sys.stderr)
        return 1
    utts = load_kaldi_set(args.dataset, None, args.ivectors)
    bs = getattr(args, 'batch_size')
    test_ds = UtteranceStream(utts, bs, test=True)
    uttids = UttIdStream(utts, bs)
    writer = kaldi_io.BaseFloatMatrixWriter(args.wxfilename)
    try:
        if which in ('lstm2', 'lstm'):
            net = build_deep_lstm(True, True, args.input_dim, args.num_layers, args.num_units,
                                  args.num_proj_units, args.output_dim, args.grad_clipping, not args.uni,
                                  args.use_layer_norm, args.ivector_dim,
                                  True if args.use_ivector_input else None, device=args.device)
            load_model(net, args.model)
            run_forward_lstm2(net, args, test_ds, uttids, writer)
        elif which == 'lstm2_tbptt':
            if not args.num_tbptt_steps:
                print('--num-tbptt-steps is required', file=sys.stderr)
                return 1
            net, tb = build_deep_lstm_tbptt(True, True, args.input_dim, args.num_layers, args.num_units,
                                            args.num_proj_units, args.output_dim, args.batch_size,
                                            args.num_tbptt_steps, args.grad_clipping, not args.uni,
                                            args.use_layer_norm, args.ivector_dim,
                                            True if args.use_ivector_input else None,
                                            backward_on_top=args.backward_on_top, device=args.device)
            load_model(net, args.model)
            run_forward_lstm2_tbptt(net, tb, args, test_ds, uttids, writer)
        elif which == 'ln_lstmp':
            net = deep_projection_ln_lstm_model(True, True, args.input_dim, args.output_dim, args.num_layers,
                                                None, args.num_units, args.grad_clipping, args.dropout,
                                                device=args.device)
            load_model(net, args.model)
            run_forward_ln_lstmp(net, args, test_ds, uttids, writer)
        else:
            net, _ = deep_projection_ivector_ln_model(True, True, True, args.input_dim, args.output_dim,
                                                      args.num_layers, args.num_conds, args.num_factors,
                                                      args.num_units, args.grad_clipping, args.dropout,
                                                      cond_dim=args.ivector_dim, device=args.device)
            load_model(net, args.model)
            run_forward_ivec_ln_lstmp(net, args, test_ds, uttids, writer)
    finally:
        writer.close()
    return 0


if __name__ == '__main__':
    sys.exit(main())

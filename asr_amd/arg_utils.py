"""Command-line flags of the reference's hot-path scripts, same names / defaults, table-driven:
  deep_lstm   libs/deep_lstm_utils.py:4-35 (+ save-path naming :42-72)
  hyperlstm   libs/hyperlstm_utils.py:3-27 (+ :34-45); plus --layer-name / --use-ivector-model, which
              scripts/exp_deep_hyperlstm.py:63,139 read but the reference parser never defines
  lhuclstm    libs/lhuclstm_utils.py:4-43
`--data-path` names a Fuel HDF5 file in the reference; here it names a Kaldi feature rspecifier or an
.npz batch file (asr_amd/data.py), since h5py/Fuel are outside the hot path.
"""
import argparse
import os

_COMMON = [
    ('--batch-size', int, 2), ('--num-layers', int, 1), ('--learn-rate', float, 0.001),
    ('--grad-clipping', float, 1.0), ('--data-path', str, '/u/songinch/song/data/speech/wsj_fbank123.h5'),
    ('--input-dim', int, 123), ('--output-dim', int, 3436), ('--ivector-dim', int, 100),
    ('--num-epochs', int, 50), ('--train-dataset', str, 'train_si284'), ('--valid-dataset', str, 'test_dev93'),
    ('--test-dataset', str, 'test_eval92'), ('--reload-model', str, None),
    ('--tmpdir', str, '/Tmp/songinch/data/speech'),
]
_COMMON_FLAGS = ['--use-ivector-input', '--no-copy', '--use-layer-norm']

_TABLE = {
    'deep_lstm': dict(
        opts=[('--num-units', int, 10), ('--num-proj-units', int, 0), ('--num-tbptt-steps', int, 0),
              ('--right-context', int, 0), ('--skip', int, 0), ('--delay', int, 0), ('--log-freq', int, 50)],
        flags=['--skip-random', '--uni', '--no-reload', '--backward-on-top']),
    'hyperlstm': dict(
        opts=[('--num-nodes', int, 10), ('--num-hyper-nodes', int, 4), ('--num-proj-nodes', int, 2),
              ('--train-disp-freq', int, 100), ('--layer-name', str, 'HyperLSTMLayer')],
        flags=['--unidirectional', '--use-ivector-model', '--no-reload']),
    'lhuclstm': dict(
        opts=[('--num-units', int, 10), ('--num-pred-layers', int, 1), ('--num-pred-units', int, 128),
              ('--num-tbptt-steps', int, 0), ('--num-seqsum-layers', int, 1), ('--num-seqsum-units', int, 128),
              ('--seqsum-output-dim', int, 100), ('--log-freq', int, 50), ('--num-proj-units', int, 0),
              ('--layer-name', str, 'IVectorLHUCLSTMLayer'), ('--mb-loss-lambda', float, 0.1)],
        flags=['--use-ivector-model', '--use-mb-loss', '--uni', '--no-reload']),
}


def get_arg_parser(family='deep_lstm'):
    parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    spec = _TABLE[family]
    for name, typ, default in _COMMON + spec['opts']:
        parser.add_argument(name, default=default, type=typ)
    for name in _COMMON_FLAGS + spec['flags']:
        parser.add_argument(name, action='store_true')
    return parser


def get_save_path(args, family='deep_lstm'):
    if family == 'hyperlstm':
        fn = './wsj_hyperlstm_lr{}_gc{}_l{}_n{}_b{}'.format(args.learn_rate, args.grad_clipping,
                                                            args.num_layers, args.num_nodes, args.batch_size)
        if args.use_ivector_input:
            fn += '_ivi{}'.format(args.ivector_dim)
        if args.unidirectional:
            fn += '_uni'
        return fn + '_hn{}_pn{}'.format(args.num_hyper_nodes, args.num_proj_nodes)
    parts = [os.path.splitext(os.path.basename(args.data_path))[0], 'lr{}'.format(args.learn_rate),
             'gc{}'.format(args.grad_clipping), 'l{}'.format(args.num_layers), 'n{}'.format(args.num_units)]
    if args.num_proj_units:
        parts.append('pjn{}'.format(args.num_proj_units))
    if args.use_ivector_input:
        parts.append('iv{}'.format(args.ivector_dim))
    parts.append('b{}'.format(args.batch_size))
    if args.use_layer_norm:
        parts.append('ln')
    if args.uni:
        parts.append('uni')
    if getattr(args, 'num_tbptt_steps', 0):
        parts.append('tb{}'.format(args.num_tbptt_steps))
    if getattr(args, 'delay', 0):
        parts.append('d{}'.format(args.delay))
    if getattr(args, 'right_context', 0):
        parts.append('rc{}'.format(args.right_context))
    if getattr(args, 'backward_on_top', False):
        parts.append('btop')
    parts.append(args.train_dataset)
    if getattr(args, 'skip', 0):
        parts.append('skip{}{}'.format('r' if args.skip_random else '', args.skip))
    return '_'.join(parts)

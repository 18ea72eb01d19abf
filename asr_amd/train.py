"""Epoch loops around the compiled callables: what the reference's hot-path experiment scripts do per
epoch, what they save and when, and the lines they print (scripts/exp_deep_lstm2.py:113-166 and the
same block in scripts/exp_deep_hyperlstm.py; scripts/exp_deep_lstm2_tbptt.py:126-200 for the windowed
form: one optimiser step PER WINDOW, recurrent state reset per batch).

One `EpochLoop` holds what both forms share (resume, status lines, model / history files, best-model
rule); the two forms differ only in how one batch is fed (`feed_whole` / `feed_windows`) and in which
evaluation function scores the held-out streams.  A stream is any object with get_epoch_iterator()
yielding the reference's 6-tuples (asr_amd.data).  `lr_schedule` is the script-level learning-rate
decay of scripts/hypernets/exp_prj_ln_lstm_fix.py:316.
"""
from __future__ import print_function

import math

from .comp_graph_utils import eval_net, eval_net_tbptt
from .utils import EvalHistory, StopWatch, gen_win, save_network, show_status


def lr_schedule(base_lr, epoch_idx):
    """lr / 1.04**max(epoch/2 - 1, 0) with the reference's Python-2 integer division of the epoch
    index (scripts/hypernets/exp_prj_ln_lstm_fix.py:316)."""
    return base_lr / (1.04 ** max(epoch_idx // 2 - 1, 0))


class EpochLoop(object):
    def __init__(self, args, network, trainer_params, feed, evaluate, train_lr=None):
        self.args = args
        self.network = network
        self.trainer_params = trainer_params
        self.feed = feed                  # batch -> (ce_frame_sum, scored frames, grads_norm)
        self.evaluate = evaluate          # stream -> (ce_frame, fer)
        self.train_lr = train_lr          # updates.SharedScalar, decayed per epoch when given
        self.base_lr = None if train_lr is None else float(train_lr.get_value())
        self.history = EvalHistory()

    def _train_epoch(self, e_idx, stream):
        args = self.args
        ce_sum = frames = 0.0
        since_status = StopWatch()
        for b_idx, batch in enumerate(stream.get_epoch_iterator(), start=1):
            ce, n, grads_norm = self.feed(batch)
            ce_sum += ce
            frames += n
            if b_idx % args.log_freq == 0:
                show_status(args.save_path, ce_sum / frames, grads_norm, b_idx, args.batch_size, e_idx)
                since_status.print_elapsed()
                since_status.reset()
        return ce_sum / frames

    def run(self, train_ds, valid_ds, test_ds, done_epochs=0):
        args = self.args
        for e_idx in range(1, args.num_epochs + 1):
            if e_idx <= done_epochs:                       # resumed run: these are in the model file
                print('Skip Epoch {}'.format(e_idx))
                continue
            if self.train_lr is not None:
                self.train_lr.set_value(lr_schedule(self.base_lr, e_idx))
            whole_epoch = StopWatch()
            print('--\nEpoch {} starts\n--'.format(e_idx))
            train_ce = self._train_epoch(e_idx, train_ds)
            print('End of Epoch {}'.format(e_idx))
            whole_epoch.print_elapsed()
            print('Saving the network')
            save_network(self.network, self.trainer_params, e_idx, args.save_path + '_last_model.pkl')
            print('Evaluating the network on the validation dataset')
            valid, test = self.evaluate(valid_ds), self.evaluate(test_ds)
            print('Train CE: {}'.format(train_ce))
            print('Valid CE: {}, FER: {}'.format(*valid))
            print('Test  CE: {}, FER: {}'.format(*test))
            if self.history.add(train_ce, valid, test):
                save_network(self.network, self.trainer_params, e_idx, args.save_path + '_best_model.pkl')
            print('Saving the evaluation history')
            self.history.save(args.save_path + '_eval_history.pkl')
        return self.history


def feed_whole(training_fn, use_ivectors):
    """One optimiser step per batch; every target frame is scored."""
    def feed(batch):
        x, xm, iv, _, y, ym = batch
        ce, grads_norm = training_fn(x, xm, iv, y, ym) if use_ivectors else training_fn(x, xm, y, ym)
        return ce, ym.sum(), grads_norm
    return feed


def feed_windows(training_fn, tbptt_layers, use_ivectors, num_tbptt_steps, right_context, delay):
    """One optimiser step per window; the reported norm is the root of the windows' squared norms."""
    def feed(batch):
        for layer in tbptt_layers:
            layer.reset(batch[0].shape[0])
        ce_sum = sq = 0.0
        for k, (x, xm, iv, _, y, ym) in enumerate(gen_win(batch, num_tbptt_steps, right_context)):
            first = 1 if k == 0 else 0
            ce, sq_k = (training_fn(x, xm, iv, y, ym, first) if use_ivectors
                        else training_fn(x, xm, y, ym, first))
            ce_sum += ce
            sq += sq_k
        return ce_sum, batch[5][:, delay:].sum(), math.sqrt(sq)
    return feed


def run_training(args, network, training_fn, trainer_params, predict_fn, train_ds, valid_ds, test_ds,
                 pretrain_total_epoch_cnt=0, train_lr=None):
    delay = getattr(args, 'delay', 0)
    loop = EpochLoop(args, network, trainer_params, feed_whole(training_fn, args.use_ivector_input),
                     lambda ds: eval_net(predict_fn, ds, args.use_ivector_input, delay=delay), train_lr)
    return loop.run(train_ds, valid_ds, test_ds, pretrain_total_epoch_cnt)


def run_training_tbptt(args, network, tbptt_layers, training_fn, trainer_params, predict_fn, train_ds,
                       valid_ds, test_ds, pretrain_total_epoch_cnt=0, train_lr=None):
    loop = EpochLoop(args, network, trainer_params,
                     feed_windows(training_fn, tbptt_layers, args.use_ivector_input, args.num_tbptt_steps,
                                  args.right_context, args.delay),
                     lambda ds: eval_net_tbptt(predict_fn, ds, tbptt_layers, args.num_tbptt_steps,
                                               args.batch_size, args.right_context, args.use_ivector_input,
                                               delay=args.delay), train_lr)
    return loop.run(train_ds, valid_ds, test_ds, pretrain_total_epoch_cnt)

"""Training loops of the reference's hot-path experiment scripts (what one epoch does, what is saved
and when), with the streams passed in:

  run_training        scripts/exp_deep_lstm2.py:113-166, scripts/exp_deep_hyperlstm.py (same skeleton)
  run_training_tbptt  scripts/exp_deep_lstm2_tbptt.py:126-200: one optimiser step PER WINDOW,
                      state reset per batch, loss on the first `context` frames of each window

A stream is any object with get_epoch_iterator() yielding the reference's 6-tuples (asr_amd.data).
"""
from __future__ import print_function

import math

from .comp_graph_utils import eval_net, eval_net_tbptt
from .utils import (EvalRecord, StopWatch, best_fer, gen_win, save_eval_history, save_network, show_status)


def _epoch_end(args, network, trainer_params, e_idx, avg_train_ce, evals, eval_history):
    (valid_ce_frame, valid_fer), (test_ce_frame, test_fer) = evals
    print('Train CE: {}'.format(avg_train_ce))
    print('Valid CE: {}, FER: {}'.format(valid_ce_frame, valid_fer))
    print('Test  CE: {}, FER: {}'.format(test_ce_frame, test_fer))
    if valid_fer < best_fer(eval_history):
        save_network(network, trainer_params, e_idx, '{}_best_model.pkl'.format(args.save_path))
    print('Saving the evaluation history')
    eval_history.append(EvalRecord(avg_train_ce, valid_ce_frame, valid_fer, test_ce_frame, test_fer))
    save_eval_history(eval_history, args.save_path + '_eval_history.pkl')


def run_training(args, network, training_fn, trainer_params, predict_fn, train_ds, valid_ds, test_ds,
                 pretrain_total_epoch_cnt=0):
    eval_history = [EvalRecord(10.1, 10.0, 1.0, 10.0, 1.0)]
    delay = getattr(args, 'delay', 0)
    for e_idx in range(1, args.num_epochs + 1):
        if e_idx <= pretrain_total_epoch_cnt:
            print('Skip Epoch {}'.format(e_idx))
            continue
        epoch_sw = StopWatch()
        print('--\nEpoch {} starts\n--'.format(e_idx))
        total_ce_sum = 0.0
        total_frame_count = 0
        status_sw = StopWatch()
        for b_idx, data in enumerate(train_ds.get_epoch_iterator(), start=1):
            input_data, input_mask, ivector_data, ivector_mask, target_data, target_mask = data
            if args.use_ivector_input:
                train_output = training_fn(input_data, input_mask, ivector_data, target_data, target_mask)
            else:
                train_output = training_fn(input_data, input_mask, target_data, target_mask)
            ce_frame_sum, network_grads_norm = train_output
            total_ce_sum += ce_frame_sum
            total_frame_count += target_mask.sum()
            if b_idx % args.log_freq == 0:
                show_status(args.save_path, total_ce_sum / total_frame_count, network_grads_norm, b_idx,
                            args.batch_size, e_idx)
                status_sw.print_elapsed(); status_sw.reset()
        print('End of Epoch {}'.format(e_idx))
        epoch_sw.print_elapsed()
        print('Saving the network')
        save_network(network, trainer_params, e_idx, args.save_path + '_last_model.pkl')
        print('Evaluating the network on the validation dataset')
        evals = [eval_net(predict_fn, ds, args.use_ivector_input, delay=delay) for ds in (valid_ds, test_ds)]
        _epoch_end(args, network, trainer_params, e_idx, total_ce_sum / total_frame_count, evals, eval_history)
    return eval_history


def run_training_tbptt(args, network, tbptt_layers, training_fn, trainer_params, predict_fn, train_ds,
                       valid_ds, test_ds, pretrain_total_epoch_cnt=0):
    eval_history = [EvalRecord(10.1, 10.0, 1.0, 10.0, 1.0)]
    for e_idx in range(1, args.num_epochs + 1):
        if e_idx <= pretrain_total_epoch_cnt:
            print('Skip Epoch {}'.format(e_idx))
            continue
        epoch_sw = StopWatch()
        print('--\nEpoch {} starts\n--'.format(e_idx))
        status_sw = StopWatch()
        total_ce_sum = 0.0
        total_frame_count = 0
        for b_idx, batch in enumerate(train_ds.get_epoch_iterator(), start=1):
            i_data, t_mask = batch[0], batch[5]
            for l in tbptt_layers:
                l.reset(i_data.shape[0])
            network_grads_norm = 0.0
            for win_idx, win in enumerate(gen_win(batch, args.num_tbptt_steps, args.right_context), start=1):
                input_data, input_mask, ivector_data, ivector_mask, target_data, target_mask = win
                is_first_win = 1 if win_idx == 1 else 0
                if args.use_ivector_input:
                    train_output = training_fn(input_data, input_mask, ivector_data, target_data, target_mask,
                                               is_first_win)
                else:
                    train_output = training_fn(input_data, input_mask, target_data, target_mask, is_first_win)
                ce_frame_sum, network_grads_norm_sum = train_output
                total_ce_sum += ce_frame_sum
                network_grads_norm += network_grads_norm_sum
            network_grads_norm = math.sqrt(network_grads_norm)
            total_frame_count += t_mask[:, args.delay:].sum()
            if b_idx % args.log_freq == 0:
                show_status(args.save_path, total_ce_sum / total_frame_count, network_grads_norm, b_idx,
                            args.batch_size, e_idx)
                status_sw.print_elapsed(); status_sw.reset()
        print('End of Epoch {}'.format(e_idx))
        epoch_sw.print_elapsed()
        print('Saving the network')
        save_network(network, trainer_params, e_idx, args.save_path + '_last_model.pkl')
        print('Evaluating the network on the validation dataset')
        evals = [eval_net_tbptt(predict_fn, ds, tbptt_layers, args.num_tbptt_steps, args.batch_size,
                                args.right_context, args.use_ivector_input, delay=args.delay)
                 for ds in (valid_ds, test_ds)]
        _epoch_end(args, network, trainer_params, e_idx, total_ce_sum / total_frame_count, evals, eval_history)
    return eval_history

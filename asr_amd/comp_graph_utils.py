"""The compiled callables of libs/comp_graph_utils.py, same names / argument order / return lists:

  trainer :68-100          training_fn(x, mask, [ivec,] y, ymask) -> [ce_frame_sum, grads_norm]
  trainer_tbptt :102-137   training_fn(x, mask, [ivec,] y, ymask, is_first_win) -> [ce_frame_sum, grads_norm**2]
  predictor :158-169       predict_fn(...)  -> [ce_frame_sum, pred_idx]
  predictor_tbptt :139-156 predict_fn(..., is_first_win) -> [ce_frame_sum, pred_idx]
  ff :321-333              ff_fn(x, mask[, ivec]) -> [posteriors (B,T,C)]
  ff_tbptt :335-351        ff_fn(x, mask, [ivec,] is_first_win) -> [posteriors]
  eval_net :171-198, eval_net_tbptt :278-319   (what they return; FER is counted on the device)

The reference builds these from Theano symbols; here the symbolic arguments (`input_data`, ... ) are
placeholders that are only tested for truthiness (`ivector_data` switches the i-vector argument on),
and the returned callables take numpy arrays (or cuda tensors), run the HIP path synchronously and
return numpy scalars / arrays, as a `theano.function` does.  With torch.distributed initialised each
rank feeds its own rows and the gradient all-reduce of asr_amd.dp runs inside training_fn.
"""
import numpy as np
import torch

from . import dp
from .utils import gen_win


def _dev(network, a, dtype=None):
    if a is None:
        return None
    if isinstance(a, torch.Tensor):
        t = a.to(network.device)
    else:
        t = torch.from_numpy(np.ascontiguousarray(a)).to(network.device)
    if dtype is not None and t.dtype != dtype:
        t = t.to(dtype)
    return t


def _given(v):
    return v is not None and v is not False


def _split_args(args, has_ivec, n_tail):
    """(x, mask, [ivec,] *tail) -> x, mask, ivec, tail"""
    need = 2 + (1 if has_ivec else 0) + n_tail
    if len(args) != need:
        raise TypeError('expected %d positional inputs, got %d' % (need, len(args)))
    x, mask = args[0], args[1]
    ivec = args[2] if has_ivec else None
    tail = args[need - n_tail:]
    return x, mask, ivec, tail


def _reducer(network):
    r = getattr(network, '_grad_allreduce', None)
    if r is None:
        r = network._grad_allreduce = dp.GradAllReduce(network)
    return r


def _loss(network, args, has_ivec, train, delay=0, context=0, tbptt=False):
    x, mask, ivec, tail = _split_args(args, has_ivec, 3 if tbptt else 2)
    y, ymask = tail[0], tail[1]
    is_first = int(tail[2]) if tbptt else 1
    d = delay if (not tbptt or is_first) else 0          # delayed_tbptt: only on the first window
    c = context if (tbptt and not delay) else 0          # `elif context` (:52-55)
    f32 = torch.float32
    return network.loss(_dev(network, x, f32), _dev(network, mask, f32), _dev(network, y, torch.int32),
                        _dev(network, ymask, f32), cond=_dev(network, ivec, f32), train=train,
                        context=c, delay=d)


def trainer(input_data, input_mask, target_data, target_mask, network, updater, learning_rate, delay=0,
            load_updater_params=None, ivector_data=None):
    has_ivec = _given(ivector_data)
    update, trainer_params = updater(loss_or_grads=None, params=network.get_all_params(trainable=True),
                                     learning_rate=learning_rate, load_params_dict=load_updater_params)
    red = _reducer(network)

    def training_fn(*args):
        ce, _, _ = _loss(network, args, has_ivec, True, delay=delay)
        network.backward()
        red.finish()
        red.agree_on_status()        # BEFORE the update: an aborted scan must not reach the parameters
        gn = update.apply(network, want_sqrt=True)
        out = torch.cat([ce, gn]).cpu().numpy()
        return [np.float32(out[0]), np.float32(out[1])]

    return training_fn, trainer_params


def trainer_tbptt(input_data, input_mask, target_data, target_mask, network, updater, learning_rate,
                  tbptt_layers, is_first_win, delay, context, load_updater_params=None, ivector_data=None):
    has_ivec = _given(ivector_data)
    update, trainer_params = updater(loss_or_grads=None, params=network.get_all_params(trainable=True),
                                     learning_rate=learning_rate, load_params_dict=load_updater_params)
    red = _reducer(network)

    def training_fn(*args):
        ce, _, _ = _loss(network, args, has_ivec, True, delay=delay, context=context, tbptt=True)
        network.backward()
        red.finish()
        red.agree_on_status()
        gn = update.apply(network, want_sqrt=False)          # un-rooted sum (:111)
        out = torch.cat([ce, gn]).cpu().numpy()
        return [np.float32(out[0]), np.float32(out[1])]

    return training_fn, trainer_params


def predictor(input_data, input_mask, target_data, target_mask, network, delay=0, ivector_data=None):
    has_ivec = _given(ivector_data)

    def fn(*args):
        ce, _, pred = _loss(network, args, has_ivec, False, delay=delay)
        network.check_status()
        return [np.float32(ce.item()), pred.cpu().numpy().astype(np.int64)]

    def stats(*args):
        ce, hits, _ = _loss(network, args, has_ivec, False, delay=delay)
        out = torch.cat([ce, hits]).cpu().numpy()
        network.check_status()
        return np.float32(out[0]), float(out[1])

    fn.stats = stats
    return fn


def predictor_tbptt(input_data, input_mask, target_data, target_mask, network, tbptt_layers, is_first_win,
                    delay, context, ivector_data=None):
    has_ivec = _given(ivector_data)

    def fn(*args):
        ce, _, pred = _loss(network, args, has_ivec, False, delay=delay, context=context, tbptt=True)
        network.check_status()
        return [np.float32(ce.item()), pred.cpu().numpy().astype(np.int64)]

    def stats(*args):
        ce, hits, _ = _loss(network, args, has_ivec, False, delay=delay, context=context, tbptt=True)
        out = torch.cat([ce, hits]).cpu().numpy()
        network.check_status()
        return np.float32(out[0]), float(out[1])

    fn.stats = stats
    return fn


def ff(network, input_data, input_mask, ivector_data=None, log=False):
    """`log=True` = the in-graph max-subtracted log-softmax of bin/forward_ln_lstmp.py:55-62."""
    has_ivec = _given(ivector_data)

    def predict_fn(*args):
        x, mask, ivec, _ = _split_args(args, has_ivec, 0)
        f32 = torch.float32
        out = network.posteriors(_dev(network, x, f32), _dev(network, mask, f32),
                                 cond=_dev(network, ivec, f32), log=log)
        network.check_status()
        return [out.cpu().numpy()]

    return predict_fn


def ff_cond(network, input_data, input_cond, input_mask, log=True):
    """bin/forward_ivec_ln_lstmp.py:58-65: argument order (input_data, input_cond, input_mask)."""
    def predict_fn(x, cond, mask):
        f32 = torch.float32
        out = network.posteriors(_dev(network, x, f32), _dev(network, mask, f32),
                                 cond=_dev(network, cond, f32), log=log)
        network.check_status()
        return [out.cpu().numpy()]

    return predict_fn


def ff_tbptt(input_data, input_mask, network, is_first_win, delay, num_tbptt_steps, ivector_data=None):
    has_ivec = _given(ivector_data)

    def fn(*args):
        x, mask, ivec, tail = _split_args(args, has_ivec, 1)
        first = int(tail[0])
        f32 = torch.float32
        o = network.posteriors(_dev(network, x, f32), _dev(network, mask, f32), cond=_dev(network, ivec, f32))
        network.check_status()
        o = o.cpu().numpy()
        if delay:
            o = o[:, delay:, :] if first else o
        elif num_tbptt_steps:
            o = o[:, :num_tbptt_steps, :]
        return [o]

    return fn


class _Tally(object):
    """Running sums of one evaluation pass: CE frame sum, correct frames, scored frames."""
    def __init__(self):
        self.ce = 0.0
        self.hits = 0.0
        self.frames = 0.0

    def result(self):
        return self.ce / self.frames, 1.0 - self.hits / self.frames


def _score(predict_fn, inputs, y, ymask):
    """-> (ce_frame_sum, correct frames or None, pred_idx or None).  The callables built by this module
    count the correct frames on the device (`row_ok` of the softmax-CE kernel summed by asr_reduce_sum)
    and hand back two scalars; the (B,T) argmax is only copied to the host for a foreign predict_fn."""
    fast = getattr(predict_fn, 'stats', None)
    if fast is not None:
        ce, hits = fast(*inputs)
        return ce, hits, None
    ce, pred = predict_fn(*inputs)
    return ce, None, pred


def eval_net(predict_fn, data_stream, use_ivectors=False, delay=0):
    """-> (CE per frame, frame error rate) over one pass of the stream; the first `delay` target frames
    of every utterance are not scored (libs/comp_graph_utils.py:18-19,188)."""
    tally = _Tally()
    for x, xm, iv, _, y, ym in data_stream.get_epoch_iterator():
        inputs = (x, xm, iv, y, ym) if use_ivectors else (x, xm, y, ym)
        ce, hits, pred = _score(predict_fn, inputs, y, ym)
        scored = ym[:, delay:]
        if hits is None:
            hits = ((y[:, delay:] == pred) * scored).sum()
        tally.ce += ce
        tally.hits += hits
        tally.frames += scored.sum()
    return tally.result()


def eval_net_skip(predict_fn, data_stream, skip, skip_random, use_ivectors=False):
    """Frame-skipped evaluation (libs/comp_graph_utils.py:200-236): the network sees every `skip`-th frame of every
    source (transformers.skip_frames: a fixed phase, or a random frame of each group of `skip`), CE per frame is taken
    over the frames it saw, and the frame error rate over ALL frames of the utterance with each prediction repeated
    `skip` times (numpy.repeat, cut to the utterance length).  (The reference indexes the repeated (B, T) prediction
    with three subscripts, :222, which numpy refuses; the two-subscript cut it means is what runs here, as in
    bin/forward_lstm2.py:96-102.)"""
    from .transformers import skip_frames, unskip
    ce_sum = ce_frames = hits = frames = 0.0
    for batch in data_stream.get_epoch_iterator():
        x, xm, iv, ivm, y, ym = [np.asarray(a) for a in batch]
        sx, sxm, siv, _, sy, sym = [np.asarray(a) for a in skip_frames((x, xm, iv, ivm, y, ym), skip, skip_random)]
        inputs = (sx, sxm, siv, sy, sym) if use_ivectors else (sx, sxm, sy, sym)
        ce, pred = predict_fn(*inputs)
        pred = unskip(np.asarray(pred), skip, y.shape[1])
        ce_sum += float(ce)
        ce_frames += float(sym.sum())
        hits += float(((y == pred) * ym).sum())
        frames += float(ym.sum())
    return ce_sum / ce_frames, 1.0 - hits / frames


def eval_net_tbptt(predict_fn, data_stream, tbptt_layers, num_tbptt_steps, batch_size, right_context,
                   use_ivectors=False, delay=0):
    """The windowed form: state reset per batch, one predict call per window (first one flagged), the
    per-window predictions laid end to end against the whole utterance's targets."""
    tally = _Tally()
    for batch in data_stream.get_epoch_iterator():
        y_all, ym_all = batch[4], batch[5]
        for layer in tbptt_layers:
            layer.reset(batch[0].shape[0])
        hits = 0.0
        preds = []
        for k, (x, xm, iv, _, y, ym) in enumerate(gen_win(batch, num_tbptt_steps, right_context)):
            first = 1 if k == 0 else 0
            inputs = (x, xm, iv, y, ym, first) if use_ivectors else (x, xm, y, ym, first)
            ce, h, pred = _score(predict_fn, inputs, y, ym)
            tally.ce += ce
            if h is None:
                preds.append(pred)
            else:
                hits += h
        scored = ym_all[:, delay:]
        if preds:
            hits = ((y_all[:, delay:] == np.concatenate(preds, axis=1)) * scored).sum()
        tally.hits += hits
        tally.frames += scored.sum()
    return tally.result()

"""Data-parallel training over the GPUs of one node: one process per GPU, torch.distributed with the
"nccl" backend (= RCCL over xGMI on ROCm).  The reference has no distributed code; the
only exchange step of the path is the SUM all-reduce of the flat fp32 gradient, bucketed per layer
in reverse topological order and launched on a side stream as soon as that layer's BPTT has produced
its gradients, so it overlaps with the BPTT of the layers below.  Batch rows never move."""
import os

import torch
import torch.distributed as dist


class GradAllReduce(object):
    def __init__(self, net, group=None):
        self.net = net
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        net.world = self.world
        net.dropout_rank = dist.get_rank(group) if dist.is_initialized() else 0   # every rank its own masks
        self.cuda = net.device.type == 'cuda'
        self.third_stream = os.environ.get('ASR_DP_STREAM', '0') == '1'
        self.stream = torch.cuda.Stream(device=net.device) if (self.cuda and self.third_stream) else None
        self.pending = []
        # ASR_DP_OVERLAP=0: no bucket leaves before the last BPTT scan has finished; finish() then reduces the whole
        # flat gradient in one collective on the main stream.  The fallback for a node on which RCCL's kernels and the
        # persistent scans do not share the chip well.  What an RCCL workgroup finds on a CU beside a scan workgroup
        # (code-object metadata of this library): up to 16 rows per GPU the scan holds one wave per
        # SIMD with 232-248 of the SIMD's 512 registers per lane and 82 KB of the 160 KB of LDS, so a 256-thread
        # workgroup of <= 264 registers and <= 78 KB fits; above 16 rows (two teams) the scan's two waves per SIMD hold
        # 448-466 registers per lane and NO further wave fits on that CU until the launch ends: disable_overlap() is
        # what bench.py calls then.  A scan that started first is never pre-empted.  Unmeasured (no multi-GPU node).
        self.overlap = os.environ.get('ASR_DP_OVERLAP', '1') != '0'
        if self.world > 1 and self.overlap:
            net.grad_hook = self.bucket_ready

    def bucket_ready(self, lo, hi):
        # Called inside Network.deferred(): the current stream is the network's side stream, behind this bucket's
        # parameter-gradient GEMMs.  The collective is issued on THAT stream (ASR_DP_STREAM=1: on a third stream behind
        # an event, the round-2 form).  With a third stream the two-process rehearsal (tools/dp_gpu_check.py, gloo)
        # stalled in finish(), both ranks waiting for a bucket's collective, in 2 of 8 and 4 of 8 runs on two boxes at
        # HIP's default of 4 hardware queues per process, in 0 of 8 with GPU_MAX_HW_QUEUES=16, and in 0 of 16 with the
        # collective on the side stream (tools/dp_repeat.sh): no more streams than hardware queues
        # is the robust form, and a bucket's all-reduce in front of the next layer's parameter-gradient GEMMs costs
        # them at most its own duration.
        g = self.net.store.g[lo:hi]
        if self.cuda and self.third_stream:
            ev = torch.cuda.Event()
            ev.record()
            with torch.cuda.stream(self.stream):
                self.stream.wait_event(ev)
                self.pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group,
                                                    async_op=True))
        else:
            self.pending.append(dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Before the optimiser step: make the main stream wait for every bucket (or, without overlap, reduce the
        whole gradient now)."""
        if self.world > 1 and not self.overlap:
            dist.all_reduce(self.net.store.g, op=dist.ReduceOp.SUM, group=self.group)
            return
        for w in self.pending:
            w.wait()
        self.pending = []
        if self.cuda and self.stream is not None:
            torch.cuda.current_stream().wait_stream(self.stream)

    def disable_overlap(self):
        """As ASR_DP_OVERLAP=0, decided by the caller (batches above 16 rows per GPU: see __init__)."""
        self.overlap = False
        self.net.grad_hook = None

    def suspend_overlap(self):
        """Gradient accumulation over micro-batches: buckets must not leave before the last micro-batch has been
        added, so backward() gets no hook and the caller reduces the accumulated gradient with reduce_all()."""
        self.net.grad_hook = None

    def reduce_all(self):
        if self.world > 1:
            dist.all_reduce(self.net.store.g, op=dist.ReduceOp.SUM, group=self.group)

    def agree_on_status(self):
        """Called between the gradient exchange and the optimiser step: one host sync, then every rank
        learns whether ANY rank's persistent scan gave up (ASR_ERR_TIMEOUT), and all of them raise
        together with parameters, optimiser state and carried TBPTT state untouched by the bad step."""
        err = None
        try:
            self.net.check_status()
        except Exception as e:                    # noqa: BLE001 -- re-raised below on every rank
            err = e
        if self.world > 1:
            flag = torch.tensor([1.0 if err is not None else 0.0], device=self.net.device)
            dist.all_reduce(flag, op=dist.ReduceOp.MAX, group=self.group)
            if err is None and flag.item() != 0.0:
                from ._lib import AsrError
                err = AsrError('a persistent scan failed on another rank; step skipped on every rank')
        if err is not None:
            raise err

    def reduce_scalars(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)
        return t


def shard_rows(lengths, rank, world):
    """Rows of the global batch that `rank` takes: the batch is sorted by length and dealt
    in contiguous blocks of B/world rows, so the padded T of a rank is set by rows of similar length.
    Returns the (sorted) row indices; every row goes to exactly one rank."""
    import numpy as np
    order = np.argsort(np.asarray(lengths), kind='stable')
    B = len(order)
    if B % world:
        raise ValueError('global batch %d is not divisible by world size %d' % (B, world))
    per = B // world
    return order[rank * per:(rank + 1) * per]


def shard_batch(batch, rank, world):
    """batch: the reference's 6-tuple (or a test-time 4-tuple); arrays sliced to this rank's rows and
    trimmed to the rank's own longest utterance."""
    import numpy as np
    mask = np.asarray(batch[1])
    rows = shard_rows(mask.sum(axis=1), rank, world)
    T = int(mask[rows].sum(axis=1).max()) if len(rows) else 0
    T = max(T, 1)
    return tuple(None if a is None else np.ascontiguousarray(np.asarray(a)[rows][:, :T]) for a in batch)

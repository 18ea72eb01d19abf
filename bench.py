#!/usr/bin/env python
"""Headline benchmark (BASELINE.json): training frames/sec of a 4x512 HyperLSTM (H=128, Z=64,
bidirectional, C=3436, B=16 per GPU, two concatenated WSJ-shaped utterances per row, grad_clipping
1.0) -- forward + BPTT + gradient all-reduce + Adam, data already resident in HBM.

  python bench.py --gpus N --steps K --warmup W
  N > 1: one rank per GPU, RCCL all-reduce of the gradients.  Under torch.distributed.run (WORLD_SIZE
  set) this process IS a rank; started plainly it spawns the N ranks itself as child processes before
  anything touches the GPU, relays rank 0's JSON line and exits non-zero if a rank fails.

Prints ONE JSON line on rank 0.  `roofline` is for the dominant kernel (the persistent HyperLSTM
BPTT scan), timed live with HIP events on the launch stream; `cpu_baseline` is the numpy oracle
(fp32) timed on a bounded sample of the same workload on the host cores (N=1 only)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_UNITS, N_HYPER, N_Z, N_LAYERS, N_OUT, N_IN, BATCH = 512, 128, 64, 4, 3436, 123, 16


def scan_bytes_per_row_dir(backward):
    """Algorithmic HBM bytes per (t,b) row and direction of the HyperLSTM scan, fp32."""
    N, H = N_UNITS, N_HYPER
    if not backward:   # read xproj, xhproj; write h, hhat, gates, c, hgates, hc, (a, d_h, d_x)
        words = 4 * N + 4 * H + N + H + 4 * N + N + 4 * H + H + 12 * N
    else:              # read gates, c(t), c(t-1), (a,d_h,d_x), xproj, hgates, hc(t), hc(t-1), dh_ext
        words = 4 * N + 2 * N + 12 * N + 4 * N + 4 * H + 2 * H + N
        words += 4 * N + 4 * H + 4 * N + 12 * N          # write dXp, dXh, dA, dD
    return 4 * words


def cpu_baseline(net, sample_len=640):      # ~10 s of host work on the GPU box's 16 cores
    """Oracle (numpy fp32, host BLAS threads) fwd + BPTT + Adam on a bounded sample: the same batch
    truncated to `sample_len` frames per row."""
    from oracle import asr_oracle as O
    from asr_amd import workloads
    b = workloads.make_batch(3, BATCH, 0, max_len=sample_len)
    vals = net.get_all_param_values()
    onet = O.OracleNetwork(net.describe(), vals, dtype=np.float32)
    m = [np.zeros_like(v) for v in onet.values]; v = [np.zeros_like(x) for x in onet.values]
    t0 = time.time()
    _, _, _, grads = onet.loss_and_grads(b['x'], b['mask'], b['y'], b['mask'])
    O.adam_step(onet.values, grads, m, v, 0)
    dt = time.time() - t0
    try:
        import threadpoolctl
        cores = max([p['num_threads'] for p in threadpoolctl.threadpool_info()] + [1])
    except Exception:
        cores = 1
    return dict(value=b['frames'] / dt, unit='frames/s', cores=int(cores), kind='port',
                sample='numpy fp32 oracle, 1 step fwd+BPTT+Adam, B=%d rows truncated to %d frames '
                       '(%d frames, %.1f s)' % (BATCH, sample_len, b['frames'], dt))


def launch_ranks(args, argv):
    """`python bench.py --gpus N` outside torch.distributed.run: start the N ranks as children.  The
    parent never initialises the GPU (no torch.cuda call), it only relays rank 0's output."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), WORLD_SIZE=str(args.gpus),
               LOCAL_WORLD_SIZE=str(args.gpus))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
    procs = []
    for r in range(args.gpus):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    # a rank that dies leaves the others inside a collective until RCCL's own timeout: watch all of them and end the
    # rest (exact PIDs) as soon as one has failed.  Rank 0 prints one short line, far below the pipe's buffer.
    while any(p.poll() is None for p in procs):
        if any(p.poll() not in (None, 0) for p in procs):
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            break
        time.sleep(0.2)
    out0 = procs[0].communicate()[0].decode()
    rcs = [p.wait() for p in procs]
    sys.stdout.write(out0)
    sys.stdout.flush()
    bad = [(r, rc) for r, rc in enumerate(rcs) if rc != 0]
    if bad:
        sys.stderr.write('bench.py: ranks failed: %r\n' % (bad,))
        return 1
    return 0


class StepClock(object):
    """Per-step durations without host syncs: one event pair per step on the launch stream (wall clock
    on the CPU self-test).  The headline `value` stays total frames / total bracketed wall time; the
    median of the per-step rates is reported beside it."""
    def __init__(self, cuda):
        self.cuda = cuda
        self.ev = []

    def start(self):
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True); e.record(); self._e0 = e
        else:
            self._e0 = time.time()

    def stop(self, frames):
        if self.cuda:
            e = torch.cuda.Event(enable_timing=True); e.record()
            self.ev.append((self._e0, e, frames))
        else:
            self.ev.append((self._e0, time.time(), frames))

    def summary(self):
        if self.cuda:
            ms = np.array([a.elapsed_time(b) for a, b, _ in self.ev])
        else:
            ms = np.array([(b - a) * 1e3 for a, b, _ in self.ev])
        fr = np.array([f for _, _, f in self.ev], dtype=np.float64)
        return dict(ms_per_step_median=float(np.median(ms)), ms_per_step_min=float(ms.min()),
                    ms_per_step_max=float(ms.max()),
                    frames_per_s_median_step=float(np.median(fr / (ms * 1e-3))))


def selftest_cpu(args, rank, world):
    """CPU rehearsal of the N-rank protocol (tests/test_host.py): gloo, no kernels.  Every rank fills a
    flat gradient of a tiny network's size, the per-layer buckets go through dp.GradAllReduce exactly
    as Network.backward() hands them over (reverse topological order), the timed region is bracketed
    as in the real run.  The printed line is marked selftest: its value measures nothing."""
    import torch.distributed as dist
    from asr_amd import dp
    from asr_amd.builders import build_deep_hyperlstm
    if os.environ.get('ASR_BENCH_SELFTEST_FAIL_RANK') == str(rank):     # tests: a rank that dies before the rendezvous
        sys.exit(3)
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('gloo', rank=rank, world_size=world)
    net = build_deep_hyperlstm('hyper', None, None, 5, 2, 8, 4, 2, 6, 1.0, True, 0, device='cpu', seed=1)
    ar = dp.GradAllReduce(net)
    assert ar.world == world == args.gpus, (ar.world, world, args.gpus)
    ranges = [net.param_range('out.')] + [net.param_range('L%d.' % i) for i in range(len(net.pairs) - 1, -1, -1)]
    clock = StepClock(False)

    def step(i):
        net.store.g.fill_(float(rank + 1))
        for lo, hi in ranges:
            if net.grad_hook:
                net.grad_hook(lo, hi)
        ar.finish()
        want = world * (world + 1) / 2.0
        assert float(net.store.g.min()) == want == float(net.store.g.max())
        return 100

    for i in range(args.warmup):
        step(i)
    if world > 1:
        dist.barrier()
    t0 = time.time(); frames = 0
    for i in range(args.steps):
        clock.start(); f = step(i); clock.stop(f); frames += f
    if world > 1:
        dist.barrier()
    dt = time.time() - t0
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64); dist.all_reduce(tt, op=dist.ReduceOp.MAX); dt = tt.item()
        ff = torch.tensor([float(frames)], dtype=torch.float64); dist.all_reduce(ff); frames = ff.item()
    if rank == 0:
        out = dict(metric='train frames/sec (node) 4x512 HyperLSTM', selftest=True, value=frames / dt,
                   unit='frames/s', n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=dt * 1e3 / args.steps, higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype='f32', data='synthetic',
                   config=dict(workload='SELFTEST (CPU, gloo, no kernels): rank protocol only',
                               per_gpu_batch=BATCH, global_batch=BATCH * world, parallelism='dp%d' % world))
        out.update(clock.summary())
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def other_config(args, rank, world, dev):
    """cfg1 / cfg2 / cfg4 / cfg5 training throughput (same timing protocol, no roofline block)."""
    import torch.distributed as dist
    from asr_amd import builders, dp, gating_hyper_nets as G, workloads
    from asr_amd.utils import gen_win
    cfg = args.cfg
    BATCH = args.batch                 # rows per GPU (module default 16)
    tb = []
    if cfg == 1:
        net = builders.build_deep_lstm(None, None, 123, 3, 256, 0, 61, 1.0, True, False, 0, device=dev, seed=1)
        name = 'cfg1: 3x256 bidirectional LSTM, C=61, TIMIT-shaped (the reference\'s CPU-runnable case)'
    elif cfg == 2:
        net = G.deep_projection_ln_lstm_model(None, None, 123, 3436, 4, None, 512, 1.0, 0.0, device=dev, seed=1)
        name = 'cfg2: 4x512 LN-LSTMP (P=256) bidirectional, C=3436, WSJ-shaped'
    elif cfg == 4:
        net, _ = G.deep_projection_ivector_ln_model(None, None, None, 123, 4174, 4, 4, 64, 512, 1.0, 0.0,
                                                    cond_dim=100, device=dev, seed=1)
        name = 'cfg4: 4x512 i-vector cond-LN LSTMP, C=4174, TED-shaped'
    else:
        net, _, tb = builders.build_deep_uni_lhuclstm_ivector_tbptt(
            'IVectorLHUCLSTMLayer', None, None, 123, 4, 512, 0, 3436, 1, 128, 0, 0, 0, BATCH, 1.0, False, False,
            100, True, device=dev, seed=1)
        name = 'cfg5: 4x512 uni LHUC-LSTM, TBPTT window 100 (one Adam step per window), lengths U{2000..6000}'
    if args.bf16:
        args.precision = 'bf16'
    if args.precision:
        net.set_precision(args.precision)       # 'bf16' is refused for the layer-norm stacks (cfg2, cfg4); 'bf16x2' is not
    ar = dp.GradAllReduce(net)
    if os.environ.get('ASR_DP_OVERLAP') is None:
        ar.disable_overlap()                    # as for cfg3 (main): one collective after the last BPTT scan by default
    batches = []
    for i in range(args.pool):
        b = workloads.make_batch(cfg, BATCH, seed_offset=1000 * rank + i)
        t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
        batches.append(dict(x=t(b['x']), m=t(b['mask']), y=t(b['y']), iv=t(b['ivec']) if 'ivec' in b else None,
                            frames=b['frames']))

    def step(i):
        b = batches[i % len(batches)]
        if args.forward_only:       # bin/forward_*.py: log-posteriors of the batch, nothing saved for BPTT
            net.posteriors(b['x'], b['m'], b['iv'], log=True)
            return b['frames']
        if cfg != 5:
            net.loss(b['x'], b['m'], b['y'], b['m'], b['iv']); net.backward(); ar.finish(); net.adam_update(1e-3)
            return b['frames']
        for l in tb:
            l.reset(BATCH)
        full = (b['x'], b['m'], b['iv'], b['m'], b['y'], b['m'])
        for xw, mw, iw, _, yw, tmw in gen_win(full, 100, 0):
            net.loss(xw, mw, yw, tmw, iw, True, 100); net.backward(); ar.finish(); net.adam_update(1e-3)
        return b['frames']

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize(); net.check_status()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    clock = StepClock(True)
    t0 = time.time(); frames = 0
    for i in range(args.steps):
        clock.start(); f = step(args.warmup + i); clock.stop(f); frames += f
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.time() - t0
    net.check_status()
    if world > 1:
        tt = torch.tensor([dt, float(frames)], dtype=torch.float64, device=dev)
        tmax = tt[0:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        fsum = tt[1:2].clone(); dist.all_reduce(fsum, op=dist.ReduceOp.SUM)
        dt = tmax.item(); frames = fsum.item()
    if rank == 0:
        if args.forward_only:
            name += ' -- FORWARD ONLY (log-posteriors, as bin/forward_*.py)'
        out = dict(metric='forward frames/sec (node)' if args.forward_only else 'train frames/sec (node)',
                   value=frames / dt, unit='frames/s', n_gpus=world,
                   steps=args.steps, warmup=args.warmup, ms_per_step=dt * 1e3 / args.steps,
                   higher_is_better=True, scaling='weak', vs_baseline=None,
                   dtype={'bf16': 'bf16-gemm/f32-scan', 'bf16x2': 'bf16x2-gemm/f32-scan'}.get(args.precision, 'f32'),
                   data='synthetic', config=dict(workload=name, per_gpu_batch=BATCH,
                                                 global_batch=BATCH * world, parallelism='dp%d' % world))
        out.update(clock.summary())
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=50)
    ap.add_argument('--warmup', type=int, default=10)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--pool', type=int, default=4, help='distinct synthetic batches per rank')
    ap.add_argument('--cfg', type=int, default=3, choices=[1, 2, 3, 4, 5],
                    help='3 = the headline workload (default); 2/4/5 = the other training configs of '
                         'BASELINE.json, reported for DESIGN.md only')
    ap.add_argument('--precision', default=None, choices=['f32x3', 'f32', 'bf16x2', 'bf16'],
                    help='how the large GEMMs multiply (asr_amd.ops): f32x3 = fp32-grade products from operands split '
                         'into three bf16 parts (default), f32 = fp32 MFMA throughout, bf16 = the optional bf16 mode')
    ap.add_argument('--bf16', action='store_true', help='same as --precision bf16: reported as dtype bf16-gemm, '
                                                        'never the default')
    ap.add_argument('--batch', type=int, default=BATCH, help='rows per GPU (weak scaling: the headline line uses 16)')
    ap.add_argument('--global-batch', type=int, default=0,
                    help='strong-scaling leg (SURVEY 8d cfg3 row, README.md:6-7,11-12: global batch 128): rows per GPU = '
                         'this / --gpus, "scaling": "strong"; the scans carry up to four 16-row chunks per workgroup')
    ap.add_argument('--micro-batch', type=int, default=32,
                    help='rows per loss()/backward() pass: more rows per GPU than this run as micro-batches of the batch '
                         'sorted by length, each padded to its own longest row, with gradient accumulation.  32 rows = '
                         'one launch per scan (two 16-row chunks per workgroup); measured at a global batch of 128 on '
                         'one GPU: 218.5 k frames/s with 32, 193.5 k with 64 (and 128 rows at once do not fit: the saved '
                         'BPTT state of cfg3 is 0.75 MB per frame row)')
    ap.add_argument('--length-sort', action='store_true',
                    help='form the global batches the way the reference\'s LengthSortTransformer does (a cache of 1024 '
                         'utterances sorted by length, cut into consecutive batches; data/transformers.py:15-63, off by '
                         'default there as here); each rank takes a contiguous block of the sorted global batch')
    ap.add_argument('--forward-only', action='store_true',
                    help='with --cfg 1/2/4: time the forward pass alone (BASELINE configs[0] is a forward-only case)')
    ap.add_argument('--selftest-cpu', action='store_true',
                    help='rehearse the N-rank protocol on the CPU over gloo (no kernels; for tests)')
    args = ap.parse_args()

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:      # before any GPU call: spawn the ranks
        sys.exit(launch_ranks(args, sys.argv[1:]))
    rank = int(os.environ.get('RANK', 0)); world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world))
    if args.selftest_cpu:
        return selftest_cpu(args, rank, world)
    torch.cuda.set_device(local)
    dev = 'cuda:%d' % local
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device(dev))

    from asr_amd.builders import build_deep_hyperlstm
    from asr_amd import workloads, dp
    if args.cfg != 3:
        return other_config(args, rank, world, dev)

    if args.global_batch:
        if args.global_batch % world:
            raise SystemExit('bench.py: --global-batch %d is not a multiple of --gpus %d' % (args.global_batch, world))
        B_local = args.global_batch // world
    else:
        B_local = args.batch
    net = build_deep_hyperlstm('hyper', None, None, N_IN, N_LAYERS, N_UNITS, N_HYPER, N_Z, N_OUT, 1.0,
                               True, 0, device=dev, seed=1234)
    if args.bf16:
        args.precision = 'bf16'
    if args.precision:
        net.set_precision(args.precision)
    from asr_amd import ops
    precision = ops.PRECISION
    ar = dp.GradAllReduce(net)
    n_micro = (B_local + args.micro_batch - 1) // args.micro_batch
    if n_micro > 1 and B_local % n_micro:
        raise SystemExit('bench.py: %d rows per GPU do not split into %d equal micro-batches' % (B_local, n_micro))
    if n_micro > 1:
        ar.suspend_overlap()        # the accumulated gradient is reduced once, after the last micro-batch
    elif B_local > 16 or os.environ.get('ASR_DP_OVERLAP') is None:
        # two teams per scan workgroup leave no registers for an RCCL wave (dp.py); and by default the bench reduces the
        # gradient in ONE collective after the last BPTT scan: RCCL kernels beside a 256-workgroup persistent scan have
        # never run on hardware (no multi-GPU node in any round), a collective that cannot get its CUs while a scan
        # waits for co-residency would end in the scans' 10 s deadline on every rank, and what the overlap can hide is
        # one all-reduce of 139 MB (~1 ms of a 175 ms step).  ASR_DP_OVERLAP=1 turns the bucketed overlap on.
        ar.disable_overlap()
    batches = []
    sorted_lens = workloads.length_sorted_batches(3, B_local * world, args.pool) if args.length_sort else None
    t_local = []                                      # padded length of every pool batch on this rank
    for i in range(args.pool):
        if sorted_lens is not None:                   # the same global batch on every rank; this rank's block of it
            b = workloads.make_batch(3, B_local, seed_offset=1000 * rank + i,
                                     lengths=sorted_lens[i][rank * B_local:(rank + 1) * B_local])
        else:
            b = workloads.make_batch(3, B_local, seed_offset=1000 * rank + i)
        parts = []
        if n_micro > 1:        # rows sorted by length (a permutation of the batch), each part padded to its own longest row
            order = np.argsort(-b['lengths'], kind='stable')
            Bm = B_local // n_micro
            for k in range(n_micro):
                rows = order[k * Bm:(k + 1) * Bm]
                Tk = int(b['lengths'][rows].max())
                parts.append(tuple(torch.from_numpy(np.ascontiguousarray(b[n][rows, :Tk])).to(dev) for n in ('x', 'mask', 'y')))
        else:
            parts.append(tuple(torch.from_numpy(b[n]).to(dev) for n in ('x', 'mask', 'y')))
        t_local.append(int(sum(p_[0].shape[1] for p_ in parts)))     # scan steps of this rank's pass(es)
        batches.append((parts, b['frames']))

    def step(i):
        parts, frames = batches[i % len(batches)]
        if len(parts) == 1:
            x, m, y = parts[0]
            net.loss(x, m, y, m)
            net.backward()
            ar.finish()
        else:
            net.world = len(parts) * world               # loss scale 1 / (rows of all parts on all ranks)
            for k, (x, m, y) in enumerate(parts):
                net.loss(x, m, y, m)
                net.backward()
                net.grad_accumulate(first=(k == 0))
            net.grad_accumulated()
            ar.reduce_all()
            net.world = world
        net.adam_update(1e-3)
        return frames

    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    net.check_status()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    net.scan_events = []
    clock = StepClock(True)
    t0 = time.time()
    frames = 0
    for i in range(args.steps):
        clock.start(); f = step(args.warmup + i); clock.stop(f); frames += f
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.time() - t0
    net.check_status()
    tt = torch.tensor([dt, float(frames)], dtype=torch.float64, device=dev)
    if world > 1:
        tmax = tt[0:1].clone(); dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        fsum = tt[1:2].clone(); dist.all_reduce(fsum, op=dist.ReduceOp.SUM)
        dt = tmax.item(); frames = fsum.item()

    # what a step waits for: every rank's scans run for its own padded length, the step ends with the slowest rank
    steps_T = torch.tensor([float(t_local[(args.warmup + i) % len(t_local)]) for i in range(args.steps)],
                           dtype=torch.float64, device=dev)
    if world > 1:
        allT = [torch.zeros_like(steps_T) for _ in range(world)]
        dist.all_gather(allT, steps_T)
        allT = torch.stack(allT).cpu().numpy()
    else:
        allT = steps_T.cpu().numpy()[None, :]
    rank_T_mean = [float(v) for v in allT.mean(axis=1)]
    rank_idle_frac = float(1.0 - allT.mean() / allT.max(axis=0).mean())
    if rank == 0:
        ev = net.scan_events
        net.scan_events = None
        bwd = [(rows, ndir, e0.elapsed_time(e1)) for k, bw, rows, ndir, e0, e1 in ev if k == 'hyper' and bw]
        fwd = [(rows, ndir, e0.elapsed_time(e1)) for k, bw, rows, ndir, e0, e1 in ev if k == 'hyper' and not bw]
        ms = float(np.mean([t for _, _, t in bwd]))
        byts = float(np.mean([rows * ndir * scan_bytes_per_row_dir(True) for rows, ndir, _ in bwd]))
        B_part = B_local // n_micro                       # rows of one loss()/backward() pass
        steps_per_launch = float(np.mean([rows / B_part for rows, _, _ in bwd]))
        ach = byts / (ms * 1e-3) / 1e9
        # HBM / fabric bytes per launch of the same kernel from the newest committed PMC passes (tools/profile_bench.sh:
        # FETCH_SIZE and WRITE_SIZE in runs of their own): 2 x FETCH + WRITE, the gfx950 correction of
        # MI355X_MICROARCH.md (FETCH_SIZE tallies wide reads at half their bytes).  The file names the library it was
        # measured on; `traffic_same_library` says whether that is the library loaded now.
        traffic = None; traffic_src = None; same_lib = None
        import glob, hashlib
        pjs = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_pmc_hyper_scan_bwd.json')))
        if pjs and B_local == BATCH:
            with open(pjs[-1]) as f:
                pm = json.load(f)
            traffic = (2.0 * pm['fetch_size_kb_per_launch'] + pm['write_size_kb_per_launch']) * 1024.0
            traffic_src = os.path.relpath(pjs[-1], ROOT)
            from asr_amd import _lib
            if pm.get('library_sha256'):
                with open(_lib.LIB_PATH, 'rb') as f:
                    same_lib = hashlib.sha256(f.read()).hexdigest() == pm['library_sha256']
        us_step = ms * 1e3 / steps_per_launch
        roof = dict(bound='hbm', kernel='hyper_scan_bwd_kernel', achieved=ach, peak=8000.0, unit='GB/s',
                    frac=ach / 8000.0, traffic=traffic, traffic_source=traffic_src, traffic_same_library=same_lib,
                    algorithmic_bytes_per_launch=byts, avg_launch_ms=ms, us_per_timestep=us_step,
                    fwd_kernel_avg_launch_ms=float(np.mean([t for _, _, t in fwd])),
                    scan_share_of_step=float(sum(t for _, _, t in bwd + fwd)) / (dt * 1e3))
        # The scan is latency-bound, not byte-bound: next to the HBM fraction, the floor that two
        # dependent chip-wide exchanges and the step's dependent MFMA chain set, from tracked measurements
        lf = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_latency_floor.json')))
        if lf:
            with open(lf[-1]) as f:
                fl = json.load(f)
            k = fl['per_step']['hyper_scan_bwd_kernel']
            floor_us = k['dependent_exchanges'] * fl['exchange_us'] + k['mfma_per_wave'] * fl['mfma_issue_ns'] * 1e-3
            launches = ((B_part + 15) // 16 + 1) // 2     # kernel launches per scan: two 16-row chunks per workgroup
            roof.update(latency_floor_us=floor_us, frac_of_latency_floor=floor_us / (us_step / launches),
                        latency_floor_source=os.path.relpath(lf[-1], ROOT),
                        latency_floor_terms=dict(exchanges_per_step=k['dependent_exchanges'], exchange_us=fl['exchange_us'],
                                                 dependent_mfma_per_wave=k['mfma_per_wave'], mfma_issue_ns=fl['mfma_issue_ns']))
        out = dict(metric='train frames/sec (node) 4x512 HyperLSTM', value=frames / dt, unit='frames/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=dt * 1e3 / args.steps,
                   higher_is_better=True, scaling='strong' if args.global_batch else 'weak', vs_baseline=None,
                   dtype={'bf16': 'bf16-gemm/f32-scan', 'bf16x2': 'bf16x2-gemm/f32-scan'}.get(precision, 'f32'), data='synthetic',
                   config=dict(workload='cfg3: 4x512 HyperLSTM (H=128,Z=64) bidirectional, C=3436, '
                                        'two concatenated WSJ-shaped utterances per row, fwd+BPTT+'
                                        'all-reduce+Adam', per_gpu_batch=B_local, global_batch=B_local * world,
                               parallelism='dp%d' % world, micro_batches=n_micro,
                               batching='length-sorted cache of 1024 utterances (LengthSortTransformer)' if args.length_sort
                               else 'independent random batches per rank (reference default: length_sort=False)',
                               scan_chunks_per_workgroup=min(2, (min(B_local, args.micro_batch) + 15) // 16),
                               allreduce='none (1 GPU)' if world == 1 else ('per-layer buckets overlapped with BPTT' if net.grad_hook is not None else 'one collective after the last BPTT scan'),
                               gemm={'f32x3': 'fp32-grade: operands split into 3 bf16 parts, 6 bf16-MFMA partial '
                                              'products, fp32 accumulate (error vs float64 below the fp32 MFMA kernel)',
                                     'f32': 'v_mfma_f32_32x32x2_f32', 'bf16': 'operands rounded to bf16',
                                     'bf16x2': 'reduced precision: operands as two bf16 parts, 3 bf16-MFMA partial products (bound: tests/test_gpu_bf16.py)'}[precision]),
                   roofline=roof)
        out.update(clock.summary())
        # padded frames per rank and step; rank_idle_frac = 1 - mean over ranks / max over ranks of them: the share of
        # a step the average rank spends waiting for the rank with the longest utterance (0 on one GPU)
        out['rank_T_mean'] = rank_T_mean
        out['rank_idle_frac'] = rank_idle_frac
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(net)
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()

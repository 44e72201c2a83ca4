#!/usr/bin/env python3
"""Headline benchmark: captions/sec, greedy decode (Captioner.forward_rl, sample_max=1) over
pre-extracted 36x2048 region features, V=10k, length 20, sentiment-word attention on
(BASELINE.json metric). One "step" = one greedy roll-out of a batch of B captions per GPU,
inputs resident in HBM, weights random-init of the reference architecture.

    python bench.py [--gpus N --steps K --warmup W --batch B --scaling weak|strong]
    python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...

Images are independent, so multi-GPU is a pure shard of the batch (no data-path collective):
weak scaling (default), value = N*B*K captions / max-over-ranks time; `--scaling strong` keeps the GLOBAL batch at B.
`--gpus N` with N > 1 and no launcher environment starts the N rank processes itself (python -m
torch.distributed.run, 127.0.0.1 rendezvous) BEFORE this process makes any GPU call, relays rank 0's one JSON line and
exits with the launcher's code; it refuses (exit 2) when the box has fewer than N GPUs, and a launcher whose WORLD_SIZE
differs from --gpus is an error - a line can never say n_gpus: 1 for a --gpus 8 request.
With N > 1 the line's `extra` carries the training curves of BASELINE configs[3] / [4]: `xe_train` (fixed 128 captions per
GPU: weak), `xe_train_strong` (fixed GLOBAL batch 1024 -> 1024/N per GPU), `rl_iteration` (Detector.forward under DP, global
B=512) and `grad_allreduce` (the 88 MB arena all-reduce alone, RCCL over xGMI).

Extra objects on the JSON line:
  roofline      dominant kernel (largest share of the decode step): achieved algorithmic
                FLOP/s (or B/s) from HIP events recorded inside the timed region
  roofline_kernels   the same for every kernel of the step (incl. the HBM-bound attention scan)
  cpu_baseline  the CPU oracle (a port of the reference, oracle/) timed on this box's host
                cores on a bounded sample of the same workload
"""
import argparse
import gc
import json
import os
import sys
import time
_T0 = time.perf_counter()

import socket
import subprocess

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

torch = np = Captioner = ops = synth = None


def load_product():
    """Imports of torch and the product, AFTER main() has decided whether this process is only a launcher."""
    global torch, np, Captioner, ops, synth
    import numpy as np_
    import torch as torch_
    from insenticap_model_amd import Captioner as Cap_, ops as ops_, synth as synth_
    torch, np, Captioner, ops, synth = torch_, np_, Cap_, ops_, synth_

PEAK_FP32_MFMA_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_F16_MFMA_TFLOPS = 2500.0   # MI355X_MICROARCH.md: dense f16 / bf16 MFMA peak
PEAK_HBM_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E spec peak (6.3 TB/s achievable)
V, R, T = 10000, 36, 20
DEFAULT_BATCH = 16384           # captions per GPU per step (7 GB of the 288 GB; rounds 1-2 ran 4096, now in the batch sweep)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=DEFAULT_BATCH, help='captions per GPU per step')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-kernel-timing', action='store_true', help='do not arm the per-kernel HIP-event timer')
    ap.add_argument('--no-extras', action='store_true', help='skip the XE-train / beam side measurements')
    ap.add_argument('--cpu-seconds', type=float, default=15.0)
    ap.add_argument('--h3-mode', type=int, default=1, choices=(0, 1, 2),
                    help='split-f16 GEMM path: 1 auto (default), 0 off = exact-fp32 MFMA tiles only, 2 force')
    ap.add_argument('--scaling', choices=('weak', 'strong'), default='weak',
                    help='weak (default): --batch captions per GPU; strong: --batch captions over all GPUs')
    ap.add_argument('--dry-ranks', type=int, default=0,
                    help='rehearse the N-rank launch path on the CPU (gloo, no GPU, no product import): the parent spawns N '
                         'ranks exactly as --gpus N does, the ranks form a group, all-reduce a token, rank 0 prints a stub line')
    ap.add_argument('--dry-fail-rank', type=int, default=-1, help='(with --dry-ranks) this rank exits non-zero: the '
                    'parent must relay the failure')
    ap.add_argument('--dry-fail-late-rank', type=int, default=-1, help='(with --dry-ranks) this rank exits non-zero AFTER '
                    'the headline, while rank 0 sits in a collective of a side measurement: rank 0 must still print its line')
    ap.add_argument('--dry-hang', action='store_true', help='(with --dry-ranks) rank 0\'s side measurements never return: '
                    'the line must go out at --extras-deadline')
    ap.add_argument('--extras-deadline', type=float, default=420.0,
                    help='under a process group: seconds the side measurements may take before rank 0 writes the line '
                         'with what it has and leaves (HeadlineGuard)')
    return ap.parse_args()


import contextlib
import functools


@contextlib.contextmanager
def no_gc():
    """Timed regions run with the cyclic collector off (after a full collection): a generation-2 pass of a
    process with torch imported costs 35-40 ms and fires at arbitrary points."""
    gc.collect()
    gc.disable()
    try:
        yield
    finally:
        gc.enable()


def device_inputs(B, seed, dev, regions=R):
    d = synth.make_inputs(B, V, synth.DEFAULT_SETTINGS, regions=regions, seq_len=T, seed=seed)
    keys = ('fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')
    return [torch.from_numpy(d[k]).to(dev) for k in keys], d


def usable_cores():
    """Host cores this process may actually use: min(affinity mask, cgroup CPU quota), capped
    at 64 (the fp32 CPU GEMMs of this workload stop scaling well before that)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as f:
            quota, period = f.read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(weights, seconds):
    """Times the oracle's greedy roll-out (same shapes, B=128 chunks) for about `seconds`."""
    from oracle import captioner_oracle as O
    cores = usable_cores()
    torch.set_num_threads(cores)
    p = O.to_params(weights)
    ids = O.Ids(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES)
    Bc = 128
    d = synth.make_inputs(Bc, V, synth.DEFAULT_SETTINGS, regions=R, seq_len=T, seed=3)
    a = [torch.from_numpy(d[k]) for k in ('fc_feats', 'att_feats', 'cpt_words', 'senti_words', 'senti_labels')]
    with torch.no_grad():
        O.forward_rl(p, ids, *[x[:8] for x in a], 2, 1)    # warm-up (thread pool, allocator)
        n, t0 = 0, time.perf_counter()
        while True:
            O.forward_rl(p, ids, *a, T, 1)
            n += 1
            el = time.perf_counter() - t0
            if el >= seconds or n >= 400:
                break
    return dict(value=round(n * Bc / el, 2), unit='captions/s', cores=cores, kind='port',
                sample='%d greedy roll-outs of B=%d (T=%d, R=%d, V=%d) by oracle/captioner_oracle.py, '
                       'torch CPU fp32, %d threads, %.1f s' % (n, Bc, T, R, V, cores, el))


def dist_on():
    return torch is not None and torch.distributed.is_available() and torch.distributed.is_initialized()


def timed_region(fn, iters, dev):
    """barrier + synchronize, `iters` calls of fn, barrier + synchronize; MAX over ranks of the elapsed seconds."""
    def fence():
        if dist_on():
            torch.distributed.barrier()
        torch.cuda.synchronize()
    fence()
    t0 = time.perf_counter()
    for _ in range(iters):
        fn()
    fence()
    el = time.perf_counter() - t0
    if dist_on():
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        el = float(tt)
    return el


def run_jobs(jobs, extra, at=None):
    """Side measurements: a failure is reported under the job's key, never fatal - EXCEPT FatalUnderGroup (not an
    Exception on purpose), which must end this rank's process: its peers are inside collectives of the same job.
    `at(key)` is told which job starts (HeadlineGuard's note)."""
    for key, fn in jobs:
        progress('extra: ' + key)
        if at is not None:
            at(key)
        try:
            extra[key] = fn()
        except Exception as e:  # noqa: BLE001
            if dist_on():
                # every job under a group holds collectives (timed_region's barriers at the least): a rank that carried on
                # alone would pair its next job's collectives with its peers' current ones
                sys.stderr.write('bench.py: side measurement %r failed under the process group: %r\n' % (key, e))
                raise FatalUnderGroup('%s: %s' % (key, repr(e)[:300])) from e
            extra[key] = {'error': repr(e)[:300]}


def progress(msg):
    """A line on stderr per leg of the run (rank 0): a long default run shows where it is (and where it died)."""
    if int(os.environ.get('RANK', '0')) == 0:
        print('[bench %6.1f s] %s' % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


class FatalUnderGroup(BaseException):
    """A failure that must END this rank's process under a process group (the other ranks are inside collectives or graph
    replays: carrying on would desynchronise them).  Deliberately not an Exception: the side-measurement loop catches
    Exception only, so this propagates to the top of the process, which exits non-zero; the launcher tears the group
    down and the parent relays the failure."""


_REAL_STDOUT = 1          # the descriptor the ONE result line goes to (main() points fd 1 at stderr while libraries talk)


class HeadlineGuard:
    """Rank 0 under a process group: the headline has been measured when the side measurements start, and it must reach
    stdout even when one of them fails or hangs ON ANY RANK.  A peer that dies makes the launcher send SIGTERM while this
    rank may sit inside a collective or a device synchronise (a Python-level handler would never run: the main thread is
    in C), and a deadlocked collective never returns at all.  So: the C-level signal handler writes to a wake-up pipe
    (signal.set_wakeup_fd), a daemon thread waits on that pipe with the side measurements' deadline as its timeout, and
    on either event it writes the line - headline plus the side measurements finished so far plus a note saying what
    happened - and ends the process non-zero (no teardown: the group is broken)."""

    def __init__(self, compose, deadline_s):
        self.compose, self.deadline_s = compose, deadline_s
        self.job = 'start'
        self._thread = self._r = self._w = self._old = self._old_fd = None

    def __enter__(self):
        import select
        import signal
        import threading
        self._r, self._w = os.pipe()
        os.set_blocking(self._w, False)
        self._old = signal.signal(signal.SIGTERM, lambda *a: None)       # (registers the C handler that feeds the pipe)
        self._old_fd = signal.set_wakeup_fd(self._w, warn_on_full_buffer=False)
        t_end = time.monotonic() + self.deadline_s

        def watch():
            while True:
                left = t_end - time.monotonic()
                ready, _, _ = select.select([self._r], [], [], max(left, 0.0))
                if ready:
                    data = os.read(self._r, 64)
                    if data == b'disarm':
                        return
                    if signal.SIGTERM not in data:
                        continue
                    why = 'SIGTERM from the launcher (another rank failed) during %r' % self.job
                else:
                    why = 'side measurements passed their deadline of %.0f s inside %r' % (self.deadline_s, self.job)
                self.emit_and_exit(why)
        self._thread = threading.Thread(target=watch, daemon=True)
        self._thread.start()
        return self

    def emit_and_exit(self, why, code=1):
        sys.stderr.write('bench.py: %s - writing the headline line and leaving\n' % why)
        os.write(_REAL_STDOUT, (self.compose(why) + '\n').encode())
        os._exit(code)

    def __exit__(self, et, ev, tb):
        import signal
        signal.set_wakeup_fd(self._old_fd if self._old_fd is not None else -1)
        signal.signal(signal.SIGTERM, self._old if self._old is not None else signal.SIG_DFL)
        os.set_blocking(self._w, True)
        os.write(self._w, b'disarm')
        self._thread.join(5)
        os.close(self._r)
        os.close(self._w)
        return False


def bench_xe_train(cap, dev, rank, world, iters=6, B=128, s2s_rows=80, curve='weak', ss_prob=0.0, regions=R):
    """BASELINE.json configs[1]/[3]: XE forward+backward+clamp+Adam on B captions per GPU plus `s2s_rows` rows of the
    seq2seq batch (train_xe.py:132-134 uses 80), train-mode dropout, scheduled sampling `ss_prob` (train_xe.py:209-212:
    0 for the first epochs, then 0.05 ... 0.25); under a process group the step takes its data-parallel form (normaliser
    all-reduce, the gradient exchange before the clamp, loss all-reduce: train_xe.py:189-192 with the exchange between
    :190 and :191).  `curve`: 'weak' = B fixed per GPU, 'strong' = the caller divided a fixed global batch by the rank
    count.  Two forms are timed: the eager step (merged unrolls; under a group: four gradient buckets reduced from inside
    the backward, dp.GradSink) and the step from HIP graphs (train_graph.XETrainGraph; under a group: one flat all-reduce
    between its two graphs); `ms_per_iter` is the faster one."""
    from insenticap_model_amd import dp
    from insenticap_model_amd.train import xe_train_step
    cap.train()
    optim, xe_crit, da_crit = cap.get_optim_criterion(4e-4)
    arena = dp.GradArena(cap.parameters()) if dist_on() else None
    # (longest caption first, as the reference's collates hand batches over: dataloader.py:17,37)
    d = synth.sort_by_length(synth.make_inputs(B, V, synth.DEFAULT_SETTINGS, regions=regions, seq_len=T, seed=500 + rank))
    s = synth.sort_by_length(synth.make_inputs(s2s_rows, V, synth.DEFAULT_SETTINGS, regions=regions, seq_len=T,
                                               seed=600 + rank))
    tt = lambda x: torch.from_numpy(x).to(dev)
    fact = (None, tt(d['fc_feats']), tt(d['att_feats']), (tt(d['captions']), d['lengths']), tt(d['cpt_words']))
    labels = tt(d['senti_labels'])
    scs = ((tt(s['captions']), s['lengths']), tt(s['cpt_words']), tt(s['senti_words']), tt(s['senti_labels']))
    from insenticap_model_amd.train_graph import XETrainGraph
    step = lambda **kw: xe_train_step(cap, optim, xe_crit, da_crit, fact, labels, scs, ss_prob, 0.1, arena=arena, **kw)
    graph = XETrainGraph(cap, optim, xe_crit, da_crit, grad_clip=0.1, arena=arena, warmup=2)
    gstep = lambda: graph.step(fact, labels, scs, ss_prob)
    graph_error = None
    exposed = None
    with no_gc():
        c0 = dp.COLLECTIVES
        for _ in range(2):
            step()
        per_iter = (dp.COLLECTIVES - c0) // 2
        el_eager = timed_region(step, iters, dev)
        if dist_on():
            # what the exchange costs on top of the compute it hides behind: the same step with the collectives of the
            # gradient buckets skipped (dp.GradSink.exchange = False: single-rank arithmetic, same launches otherwise),
            # and the flat form (one all-reduce of the whole arena after the backward)
            sink = cap.__dict__.get('_dp_sink')
            if sink is not None:
                n2 = max(iters, 12)      # (differences of ~0.1 ms: longer regions, the bucketed form timed again LAST so
                                         # that warm-up order does not read as exchange cost)
                sink.exchange = False
                try:
                    step()
                    el_dry = timed_region(step, n2, dev) / n2
                finally:
                    sink.exchange = True
                step(bucketed=False)
                el_flat = timed_region(lambda: step(bucketed=False), n2, dev) / n2
                step()
                el_buck = min(timed_region(step, n2, dev) / n2, el_eager / iters)
                exposed = dict(bucketed_ms=round((el_buck - el_dry) * 1e3, 3),
                               flat_ms=round((el_flat - el_dry) * 1e3, 3),
                               no_exchange_ms_per_iter=round(el_dry * 1e3, 2),
                               flat_ms_per_iter=round(el_flat * 1e3, 2))
                el_eager = min(el_eager, el_buck * iters)
        try:
            for _ in range(4):          # two eager steps on the graph's own streams, the capture, two replays
                gstep()
            r0 = graph.replays
            el = timed_region(gstep, iters, dev)
            replayed = graph.replays - r0
        except Exception as e:          # noqa: BLE001 - the eager figure above stands on its own
            if dist_on():
                # under a process group a failed capture is not recoverable in place (the other ranks are inside a
                # collective or a replay; a backend watchdog abort is not even an exception): this rank's process ends
                # non-zero (FatalUnderGroup is not an Exception: the side-measurement loop cannot swallow it), the
                # launcher tears the group down and the parent relays the failure - never an eager line that pretends
                # to be the N-rank measurement, never a re-exec of this GPU-initialised process
                sys.stderr.write('bench.py: graph capture failed under the process group: %r\n' % (e,))
                raise FatalUnderGroup(repr(e)[:300])
            graph_error, el, replayed = repr(e)[:300], el_eager, 0
        # third form, large batches without a group: the eager step with one chain per unroll and the RAGGED unroll
        # (Captioner.row_counts: step t runs on the rows whose caption has not ended, the classifier block over the
        # sum(lengths) rows inside their captions; same loss, same gradients - tests/test_gpu_ragged.py).  Not
        # graph-servable: the row counts are baked into the launches.
        el_ragged = None
        if not dist_on() and B >= 512 and ss_prob == 0.0:
            keep = (getattr(cap, 'pair_unrolls', None), getattr(cap, 'ragged_unroll', False))
            cap.pair_unrolls, cap.ragged_unroll = False, True
            try:
                for _ in range(3):
                    step()
                n3 = max(iters, 10)     # (4 iterations after a change of form read 5 % high: allocator and plan warm-up)
                el_ragged = timed_region(step, n3, dev) * iters / n3
            except Exception as e:      # noqa: BLE001 - the other two forms stand on their own
                graph_error = (graph_error or '') + ' ragged: ' + repr(e)[:200]
            finally:
                cap.pair_unrolls, cap.ragged_unroll = keep
    cap.eval()
    for q in cap.parameters():          # the arena's views must not outlive this measurement
        q.grad = None
    cap.__dict__.pop('_dp_sink', None)
    best = min(el, el_eager)
    out = dict(curve=curve, iters=iters, batch_per_gpu=B, global_batch=world * B, seq2seq_rows_per_gpu=s2s_rows,
               ss_prob=ss_prob, regions=regions,
               ms_per_iter=round(best / iters * 1e3, 2), captions_per_s=round(world * B * iters / best, 1),
               served_from=('eager step, merged unrolls (autograd_pair)' + ('; gradient exchange in 4 buckets from inside '
                            'the backward (dp.GradSink)' if dist_on() else '')) if el_eager <= el else
                           ('HIP graphs (train_graph.XETrainGraph: forward+backward and clamp+Adam replayed, '
                            'collectives between them); %d of %d timed iterations were replays' % (replayed, iters)),
               eager_ms_per_iter=round(el_eager / iters * 1e3, 2), graph_ms_per_iter=round(el / iters * 1e3, 2),
               grad_allreduce_mb=round(arena.nbytes / 1e6, 2) if arena else 0.0, all_reduces_per_iter=per_iter)
    out['active_positions'] = round(sum(d['lengths']) / float(T * B), 3)
    if el_ragged is not None:
        out['ragged_eager_ms_per_iter'] = round(el_ragged / iters * 1e3, 2)
        if el_ragged < best:
            out['ms_per_iter'] = out['ragged_eager_ms_per_iter']
            out['captions_per_s'] = round(world * B * iters / el_ragged, 1)
            out['served_from'] = ('eager step, one chain per unroll, ragged unroll (captioner.ragged_unroll = True: per-step '
                                  'launches and the classifier block on the positions inside the captions only; batches '
                                  'sorted by length as the reference\'s collates sort them)')
    if exposed is not None:
        out['exposed_allreduce_ms'] = exposed['bucketed_ms']
        out['exchange'] = exposed
    if graph_error is not None:
        out.update(graph_error=graph_error)
    return out


def bench_xe_exchange_one_rank(timeout=240):
    """N=1 without a launcher has no process group, so `xe_train` cannot time the gradient exchange.  A CHILD process
    (never an exec of this GPU-initialised one) runs the same eager merged XE step (B=128+80) under a ONE-RANK RCCL group:
    bucketed exchange (dp.GradSink), the buckets with their collectives skipped, and one flat all-reduce.  With one rank
    nothing crosses a link: the figure is the fixed cost of issuing and waiting for the collectives."""
    import subprocess
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29000 + os.getpid() % 2000))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'profile_xe_dp.py'), '16', '128', '--json'],
                       env=env, capture_output=True, text=True, timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError('one-rank child rc %d: %s' % (r.returncode, r.stderr[-200:]))
    out = json.loads(r.stdout.strip().splitlines()[-1])
    out['note'] = 'one-rank RCCL group in a child process: the fixed cost of the collectives, nothing to transfer'
    return out


def bench_grad_allreduce(cap, dev, world, reps=10):
    """The gradient exchange alone: one sum-all-reduce of the flat 22 063 379-float arena (RCCL over xGMI), nothing
    to overlap it with (no gradient is final before BPTT ends) - this is the exposed time inside every DP iteration."""
    n = sum(q.numel() for q in cap.parameters())
    flat = torch.zeros(n, dtype=torch.float32, device=dev)
    for _ in range(3):
        torch.distributed.all_reduce(flat)
    el = timed_region(lambda: torch.distributed.all_reduce(flat), reps, dev)
    ms = el / reps * 1e3
    return dict(mb=round(n * 4 / 1e6, 2), ranks=world, ms=round(ms, 3), backend=torch.distributed.get_backend(),
                bus_gb_per_s=round(2.0 * (world - 1) / max(world, 1) * n * 4 / (ms * 1e-3) / 1e9, 1))


def bench_small_batches(cap, dev, batches=(4, 128, 512)):
    """Latency regime of the headline path: greedy roll-outs (prologue + 20 steps, no host sync inside) at few captions
    per call.  ms_per_rollout: 10 roll-outs enqueued back to back (the headline's timing form); single_call_ms: one
    call bracketed by synchronisation (median of 10), which also exposes the host's enqueue time."""
    out = {}
    with torch.no_grad(), no_gc():
        for B in batches:
            inputs, _ = device_inputs(B, 700 + B, dev)
            for _ in range(3):
                cap(*inputs, T, 1, mode='rl')
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(10):
                cap(*inputs, T, 1, mode='rl')
            torch.cuda.synchronize()
            loop = (time.perf_counter() - t0) / 10
            ts = []
            for i in range(10):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                cap(*inputs, T, 1, mode='rl')
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t0)
            ts.sort()
            out[str(B)] = dict(ms_per_rollout=round(loop * 1e3, 3), captions_per_s=round(B / loop, 1),
                               single_call_ms=round(ts[len(ts) // 2] * 1e3, 3))
    return out


def bench_batch_sweep(cap, dev, batches=(4096, 8192)):
    """The headline workload at larger batches per step (same weights, same path): captions/s of 4 roll-outs."""
    out = {}
    with torch.no_grad(), no_gc():
        for B in batches:
            inputs, _ = device_inputs(B, 800 + B, dev)
            for _ in range(2):                       # the caching allocator settles on the larger blocks
                cap(*inputs, T, 1, mode='rl')
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(4):
                cap(*inputs, T, 1, mode='rl')
            torch.cuda.synchronize()
            el = (time.perf_counter() - t0) / 4
            out[str(B)] = dict(ms_per_rollout=round(el * 1e3, 2), captions_per_s=round(B / el, 1))
            del inputs
    return out


def bench_exact_fp32(cap, inputs, B, reps=3):
    """The same B=4096 roll-out with the split-f16 engine off (isc_set_h3_mode(0)): every GEMM on the exact-fp32 MFMA
    tiles (v_mfma_f32_32x32x2_f32, peak 157.3 TFLOP/s)."""
    prev = ops.set_h3_mode(0)
    try:
        with torch.no_grad(), no_gc():
            cap(*inputs, T, 1, mode='rl')
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                cap(*inputs, T, 1, mode='rl')
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
    finally:
        ops.set_h3_mode(prev)
    return dict(captions_per_s=round(B * reps / el, 1), ms_per_rollout=round(el / reps * 1e3, 2),
                engine='v_mfma_f32_32x32x2_f32 only (--h3-mode 0)')


def bench_table_build(cap, inputs):
    """Cost of what the timed roll-outs amortise: the [V,4H] token table relu(Emb) W_x^T and the two [V,512] sentiment-
    word tables are functions of the (frozen) weights, built by the first roll-out after a weight change and reused
    by every later one; one-off, outside the timed region."""
    with torch.no_grad(), no_gc():
        cap._tab_cache = cap._senti_tab_cache = None
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        p = cap._p()
        with ops.h3_weights_scope(cap._dev):
            cap._embedding_table(p, build=True)
            cap._senti_tables(p)
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
    return dict(ms=round(el * 1e3, 2), gflop=round(2.0 * V * 512 * (2048 + 512) / 1e9, 1))


def bench_scan_sweep(dev, batches=(128, 512, 1024, 2048, 4096), R=R):
    """the attention scan's algorithmic GB/s at the config batch sizes as well as where its rows
    (B x 192 512 B for content + sentiment) exceed the 256 MB last-level cache.  Isolated kernel, HIP events."""
    out = {}
    for B in batches:
        g = torch.Generator(device='cpu').manual_seed(B)
        mk = lambda *s: torch.rand(*s, generator=g).to(dev)
        Pc, Vc, Pw, Vw = mk(B, R, 512), mk(B, R, 512), mk(B, 11, 512), mk(B, 11, 512)
        q, q2, w, wb = mk(B, 512), mk(B, 512), mk(512), mk(1)
        oc, ow = torch.empty(B, 512, device=dev), torch.empty(B, 512, device=dev)
        pr = [ops.scan_problem(Pc, Vc, q, w, wb, oc), ops.scan_problem(Pw, Vw, q, w, wb, ow, q2=q2)]
        for _ in range(3):
            ops.attn_scan_fwd(pr, B)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 20
        e0.record()
        for _ in range(reps):
            ops.attn_scan_fwd(pr, B)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        row_bytes = (2 * R + 2 * 11) * 512 * 4.0          # 2 R E 4 content + 2 M W 4 sentiment bytes per caption
        out[str(B)] = dict(us=round(us, 1), gb_per_s=round(B * row_bytes / us / 1e3, 1),
                           frac_of_8tbs=round(B * row_bytes / us / 1e3 / 8000.0, 3),
                           row_bytes_mb=round(B * row_bytes / 1e6, 1))
    return out


def bench_epoch_loops(dev, n_img_xe=512, n_img_rl=1024):
    """A training EPOCH the way the reference's trainers run it (train_xe.py:132-192, train_rl.py:232-242): the package's
    loaders over synthetic images (captions of different lengths, 6 x 6 x 2048 regions) feeding one iteration per batch -
    loader, hand-over to the device and the iteration together.  `resident`: the features in HBM (data.DeviceFeatureStore:
    a batch is an index_select on the device); `host`: dict-backed stores + DevicePrefetcher (XE: dedup collate)."""
    from insenticap_model_amd import Captioner, Detector, data
    from insenticap_model_amd.train_graph import XETrainGraph
    import warnings
    import numpy as np
    rng = np.random.default_rng(7)
    st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
    n_img = max(n_img_xe, n_img_rl)
    fns = ['img%05d' % i for i in range(n_img)]
    fc = {fn: rng.standard_normal(2048, dtype=np.float32) * 0.5 for fn in fns}
    att = {fn: rng.standard_normal((6, 6, 2048), dtype=np.float32) * 0.5 for fn in fns}

    def caption():
        return [1] + rng.integers(4, V, size=int(rng.integers(6, T))).tolist() + [2]
    caps = {fn: [caption() for _ in range(4)] for fn in fns}
    cpts = {fn: rng.integers(4, V, size=5).tolist() for fn in fns}
    sentis = {fn: rng.integers(4, V, size=10).tolist() for fn in fns}
    scs_rows = [(caption(), rng.integers(4, V, size=5).tolist(), rng.integers(4, V, size=10).tolist(), int(rng.integers(0, 3)))
                for _ in range(80 * 16)]
    dfc = data.DeviceFeatureStore.from_arrays(fns, [fc[f] for f in fns], dev)
    datt = data.DeviceFeatureStore.from_arrays(fns, [att[f] for f in fns], dev)
    data.freeze_host_objects()          # (the caption tables: millions of small objects the cyclic collector would re-walk)
    out = {}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        # ---- XE: 32 images x 4 captions = 128 rows + 80 seq2seq rows per iteration, graph-served step
        xe_caps = {fn: caps[fn] for fn in fns[:n_img_xe]}
        for tag, a, b, kw in (('resident', dfc, datt, {}), ('host', fc, att, dict(dedup=True))):
            cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
            cap.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, synth.DEFAULT_SETTINGS).items()})
            cap.to(dev).train()
            optim, xc, dc = cap.get_optim_criterion(4e-4)
            cl = data.get_caption_dataloader(a, b, xe_caps, cpts, 0, T, 5, 32, shuffle=True, caption_width='full', **kw)
            sl = data.get_senti_corpus_with_sentis_dataloader(scs_rows, 0, T, 5, 10, 80, shuffle=True, caption_width='full')
            g = XETrainGraph(cap, optim, xc, dc, grad_clip=0.1, warmup=2)
            n = 0
            for ep in range(2):                      # the first epoch warms up and captures, the second is timed
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                src = (cl, sl) if tag == 'resident' else (data.DevicePrefetcher(cl, dev), data.DevicePrefetcher(sl, dev))
                for fact, scs in zip(*src):
                    labels = torch.zeros(fact[1].shape[0], dtype=torch.int64, device=dev)
                    # (resident: the features come as RowGather over the store; captions, concepts and the seq2seq batch
                    # are a few KB of host tensors that the graph object stages itself)
                    fact = tuple(x.to(dev) if isinstance(x, data.RowGather) else x for x in fact)
                    g.step(fact, labels, scs, 0.0)
                    n += 1
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
            out['xe128_' + tag + '_ms_per_iter'] = round(el / len(cl) * 1e3, 2)
            del g, cap
        # ---- RL: 512 images per iteration through Detector.forward (graph-served), image sentiments cached after epoch 1
        rl_caps = {fn: caps[fn] for fn in fns[:n_img_rl]}
        det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
        det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
        det.to(dev)
        det.set_ciderd_scorer({'train': rl_caps})
        for tag, a, b in (('resident', dfc, datt), ('host', fc, att)):
            fl = data.get_rl_fact_dataloader(a, b, rl_caps, cpts, sentis, 0, T, 5, 10, 512, shuffle=True, caption_width='full')
            sl = data.get_senti_corpus_with_sentis_dataloader(scs_rows, 0, T, 5, 10, 80, shuffle=True, caption_width='full')
            for ep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                det((fl, sl), 'fact', True)
                torch.cuda.synchronize()
                el = time.perf_counter() - t0
            out['rl512_' + tag + '_ms_per_iter'] = round(el / len(fl) * 1e3, 1)
    out['note'] = ('loader + hand-over + iteration per batch; resident = data.DeviceFeatureStore (features in HBM), host = dict '
                   'stores (+ DevicePrefetcher, dedup collate for XE); %d / %d images' % (n_img_xe, n_img_rl))
    return out


def bench_r196(cap, dev):
    """The reference encoder's own feature geometry: 14 x 14 = 196 regions per image (models/encoder.py:53; BASELINE.json
    quotes the 36-region bottom-up features).  Greedy decode at B = 4096, the attention scan alone (algorithmic bytes
    2 R E 4 + 2 M W 4 = 847 872 B per caption and step), an XE training iteration at B = 128 + 80, beam-5 one image."""
    out = {}
    Rg = 196
    with torch.no_grad(), no_gc():
        inputs, _ = device_inputs(4096, 900, dev, regions=Rg)
        for _ in range(2):
            cap(*inputs, T, 1, mode='rl')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            cap(*inputs, T, 1, mode='rl')
        torch.cuda.synchronize()
        el = (time.perf_counter() - t0) / 3
        out['greedy_B4096'] = dict(ms_per_rollout=round(el * 1e3, 2), captions_per_s=round(4096 / el, 1))
        beam_in = [x[:16] for x in inputs]
        del inputs
    out['scan'] = bench_scan_sweep(dev, batches=(128, 1024, 4096), R=Rg)
    try:
        b = bench_beam(cap, beam_in, n_img=16, beam=5)
        out['beam5'] = {k: b[k] for k in ('per_image_p50_ms', 'per_step_p50_us', 'full_search_p50_ms')}
    except Exception as e:  # noqa: BLE001
        out['beam5'] = {'error': repr(e)[:200]}
    x = bench_xe_train(cap, dev, 0, 1, iters=4, B=128, regions=Rg)
    out['xe_train_B128'] = {k: x[k] for k in ('ms_per_iter', 'eager_ms_per_iter', 'graph_ms_per_iter', 'served_from')}
    return out


def bench_rl(dev, iters=8, B=512, cache_image_sentiments=True, rank=0, world=1):
    """BASELINE.json configs[4]: self-critical RL iteration (Detector.forward, training=True): sampled +
    greedy roll-out per image, CIDEr-D + classifier rewards, XE (ss 0.5) + seq2seq (ss 0.25) passes,
    backward, clamp, Adam; GLOBAL B=512, T=20, 6x6x2048 grid for the sentiment detector, 5 GT captions/image.
    Under a process group: Detector.enable_data_parallel, rank r takes rows [r*B/N, (r+1)*B/N) of the fact batch and
    of the 80-row seq2seq batch; document frequencies are built from ALL images on every rank (models/decoder.py:
    163-167 with the gradient all-reduce between :165 and :166)."""
    from insenticap_model_amd import Detector, dp, rewards
    st = dict(synth.DEFAULT_SETTINGS, **synth.HELPER_SETTINGS)
    det = Detector(synth.make_idx2word(V), T, synth.SENTIMENT_CATEGORIES, {'cap_lr': 4e-5}, st)
    det.captioner.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_weights(V, st, seed=0).items()})
    det.to(dev)
    if dist_on():
        det.enable_data_parallel()
    det.cache_image_sentiments = cache_image_sentiments      # False: the frozen conv net runs on every image, every iteration
    batches, split = synth.make_rl_batches(1, B, V, st, grid=(6, 6), seq_len=T, seed=90)
    det.set_ciderd_scorer(split)
    tt = torch.from_numpy
    b = batches[0]
    lo, hi = dp.shard(B, rank, world, drop_last=True)
    fns = b[0][lo:hi]
    fact = [(fns, tt(b[1][lo:hi]).to(dev), tt(b[2][lo:hi]).to(dev), (tt(b[3][0][lo:hi]).to(dev), b[3][1][lo:hi]),
             tt(b[4][lo:hi]).to(dev), tt(b[5][lo:hi]).to(dev), {fn: b[6][fn] for fn in fns})]
    s = synth.make_inputs(80, V, st, regions=R, seq_len=T, seed=91)
    s_lo, s_hi = dp.shard(80, rank, world, drop_last=True)
    scs = [((tt(s['captions'][s_lo:s_hi]).to(dev), s['lengths'][s_lo:s_hi]), tt(s['cpt_words'][s_lo:s_hi]).to(dev),
            tt(s['senti_words'][s_lo:s_hi]).to(dev), tt(s['senti_labels'][s_lo:s_hi]).to(dev))]
    cider_t = [0.0]
    orig = rewards.get_self_critical_reward

    def timed(*a, **k):
        t0 = time.perf_counter()
        r = orig(*a, **k)
        cider_t[0] += time.perf_counter() - t0
        return r
    import insenticap_model_amd.detector as dmod
    dmod.get_self_critical_reward = timed
    orig_half = rewards.self_critical_scores          # (the graph-served iteration scores the two token matrices separately
                                                      # and resolves the function from the module per call)
    def timed_half(*a, **k):
        t0 = time.perf_counter()
        r = orig_half(*a, **k)
        cider_t[0] += time.perf_counter() - t0
        return r
    rewards.self_critical_scores = timed_half
    losses = {}

    def one():
        losses.update(det((fact, scs), 'fact', True))
    try:
        for _ in range(4):                        # warm-up: two eager iterations, the graph capture, one replay
            one()
        torch.cuda.synchronize()
        cider_t[0] = 0.0
        with no_gc():
            el = timed_region(one, iters, dev)
    finally:
        dmod.get_self_critical_reward = orig
        rewards.self_critical_scores = orig_half
    return dict(iters=iters, global_batch=(hi - lo) * world, batch_per_gpu=hi - lo, seq2seq_rows_per_gpu=s_hi - s_lo,
                ms_per_iter=round(el / iters * 1e3, 1), images_per_s=round((hi - lo) * world * iters / el, 1),
                image_sentiment_cache=bool(cache_image_sentiments),
                served_from=('HIP graphs (train_graph.RLTrainGraph: %d replays, %d eager iterations so far)'
                             % (det._rl_graph.replays, det._rl_graph.eager_steps)) if det._rl_graph is not None else 'eager',
                cider_ms_per_iter=round(cider_t[0] / iters * 1e3, 1), cider_threads=det.ciderd_scorer.n_threads,
                grad_arena_all_reduces=det.dp_arena.collectives if det.dp_arena is not None else 0,
                losses={k: round(float(v), 4) for k, v in losses.items()})


def bench_beam(cap, inputs, n_img=64, beam=5):
    """BASELINE.json configs[2]: beam 5, sentiment attention on. Reference API (one image per call)
    latency and the batched path's throughput, as Captioner.sample serves them by default (from HIP graphs once a search
    geometry has been seen twice); `eager`: the same calls with enable_beam_graphs(False)."""
    fc, att, _, sw, lab = [x[:n_img] for x in inputs]

    def single_image(warm):
        lat, per_step = [], []
        for _ in range(warm):
            cap.sample(fc[0], att[0], sw[0], lab[0:1], beam, 1, T)
        for i in range(16):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cap.sample(fc[i], att[i], sw[i], lab[i:i + 1], beam, 1, T)
            torch.cuda.synchronize()
            lat.append(time.perf_counter() - t0)
            per_step.append(lat[-1] / max(1, cap.last_beam_steps))
        lat.sort()
        per_step.sort()
        return lat, per_step

    def forced_full(warm):
        # the same search forced through all T steps: with random-init weights captions end early, a trained model's
        # often do not.  No candidate can end when <EOS> is an id the vocabulary does not contain.
        eos = cap.eos_id
        cap.eos_id = -7
        try:
            full, _ = single_image(warm)
            assert cap.last_beam_steps == T
        finally:
            cap.eos_id = eos
        return full

    # Default behaviour of Captioner.sample (round 3): a search geometry seen twice is captured into HIP graphs and replayed.
    # `eager`: the same calls with the graphs switched off (enable_beam_graphs(False)).
    with torch.no_grad(), no_gc():
        cap.enable_beam_graphs(True)
        g_lat, g_step = single_image(3)            # first call eager, second captures, then replays
        g_full = forced_full(3)
        cap.sample_batch(fc, att, sw, lab, beam, 1, T)
        cap.sample_batch(fc, att, sw, lab, beam, 1, T)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cap.sample_batch(fc, att, sw, lab, beam, 1, T)
        torch.cuda.synchronize()
        g_el = time.perf_counter() - t0
        cap.enable_beam_graphs(False)
        try:
            lat, per_step = single_image(1)
            cap.sample_batch(fc, att, sw, lab, beam, 1, T)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            cap.sample_batch(fc, att, sw, lab, beam, 1, T)
            torch.cuda.synchronize()
            el = time.perf_counter() - t0
            full = forced_full(1)
        finally:
            cap.enable_beam_graphs(True)
    # latency is bimodal with random-init weights: captions either end after ~9 steps or run all T=20;
    # per_step_p50_us (latency / executed decode steps) is the number to compare between runs
    pct = lambda xs, q: round(xs[min(len(xs) - 1, max(0, int(len(xs) * q) - (1 if q > 0.5 else 0)))] * 1e3, 2)
    return dict(beam=beam, serving='HIP graphs (default: captured on the second call of a geometry)',
                per_image_p50_ms=pct(g_lat, 0.5), per_image_p95_ms=pct(g_lat, 0.95),
                per_image_min_ms=round(g_lat[0] * 1e3, 2), per_image_max_ms=round(g_lat[-1] * 1e3, 2),
                per_step_p50_us=round(g_step[len(g_step) // 2] * 1e6, 1),
                full_search_steps=T, full_search_p50_ms=pct(g_full, 0.5), full_search_p95_ms=pct(g_full, 0.95),
                batched_images=n_img, batched_images_per_s=round(n_img / g_el, 1),
                eager=dict(per_image_p50_ms=pct(lat, 0.5), per_image_p95_ms=pct(lat, 0.95),
                           per_step_p50_us=round(per_step[len(per_step) // 2] * 1e6, 1),
                           full_search_p50_ms=pct(full, 0.5), full_search_p95_ms=pct(full, 0.95),
                           batched_images_per_s=round(n_img / el, 1)))


PMC_SUMMARY = os.path.join(ROOT, 'profiles', 'r05_b_pmc_summary_B16384.json')
# bench kernel label -> device symbols it may run as (first one present in the PMC summary wins)
KERNEL_SYMBOL = {'vocab[': ['void gemm_h3_kernel<2, false>', 'void gemm_h3_kernel<2>', 'void gemm_ld_kernel<2>', 'void gemm_kernel<4, 1, 4, 2, false, false>'],
                 'lstm[': ['void gemm_h3x_kernel<1, false>', 'void gemm_h3x_kernel<1>', 'void gemm_h3_kernel<1>', 'void gemm_ld_kernel<1>', 'void gemm_kernel<4, 1, 4, 1, false, false>', 'void gemm_xl_kernel<1>'],
                 'attn_scan[': ['void attn_scan_kernel<2, true>', 'void attn_scan_kernel<2, false>', 'void attn_scan_kernel<2>'], 'gate_mix[': ['gate_mix_kernel'],
                 'rollout_finalize[': ['rollout_finalize_wide_kernel', 'rollout_finalize_kernel']}


def pmc_traffic(name, batch):
    """HBM-side bytes per launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 gfx950
    correction + WRITE_SIZE); only valid for the batch size the counters were collected at."""
    try:
        d = json.load(open(PMC_SUMMARY))
    except (OSError, ValueError):
        return None
    if d.get('batch_per_gpu') != batch:
        return None
    for prefix, syms in KERNEL_SYMBOL.items():
        if name.startswith(prefix):
            for sym in syms:
                if sym in d['kernels']:
                    if sym in ('void gemm_h3_kernel<1>', 'void gemm_h3x_kernel<1>', 'void gemm_h3x_kernel<1, false>'):   # both LSTM cells: no per-shape figure
                        return None
                    return d['kernels'][sym]['traffic_bytes']
    return None


def pmc_counters(name, batch):
    """Matrix-pipe counters of the kernel behind a bench label, from the same committed PMC summary (separate
    rocprofv3 --pmc passes, tools/profile_round.sh): mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x
    GRBM_GUI_ACTIVE / 8), the clock the chip held, the L2 hit rate.  Per symbol (both LSTM cells share one)."""
    try:
        d = json.load(open(PMC_SUMMARY))
    except (OSError, ValueError):
        return None
    if d.get('batch_per_gpu') != batch:
        return None
    for prefix, syms in KERNEL_SYMBOL.items():
        if name.startswith(prefix):
            for sym in syms:
                k = d['kernels'].get(sym)
                if k and 'mfma_util' in k:
                    return {'symbol': sym, 'mfma_util': k['mfma_util'], 'clock_ghz': k.get('clock_ghz'),
                            'l2_hit': k.get('l2_hit'), 'source': os.path.basename(PMC_SUMMARY)}
    return None


def roofline_entry(name, rec):
    ms = rec['avg_ms']
    if rec['flops'] > 0 and rec.get('h3'):
        # split-f16 path: three f16 MFMA products per fp32 product, so the ceiling for ALGORITHMIC (fp32-equivalent)
        # FLOP/s is a third of the dense f16 peak; avg_us includes the operand-split kernel in front of the GEMM
        ach = rec['flops'] / (ms * 1e-3) / 1e12
        peak = round(PEAK_F16_MFMA_TFLOPS / 3, 1)
        return dict(kernel=name, bound='mfma', achieved=round(ach, 3), peak=peak, unit='TFLOP/s',
                    frac=round(ach / peak, 4), traffic=None, avg_us=round(ms * 1e3, 2), launches_timed=rec['n'],
                    mfma='f16 x3 split products, fp32 accumulate (peak = dense f16 2500 / 3)',
                    x_fp32_mfma_peak=round(ach / PEAK_FP32_MFMA_TFLOPS, 3))
    if rec['flops'] > 0:
        ach = rec['flops'] / (ms * 1e-3) / 1e12
        return dict(kernel=name, bound='mfma', achieved=round(ach, 3), peak=PEAK_FP32_MFMA_TFLOPS,
                    unit='TFLOP/s', frac=round(ach / PEAK_FP32_MFMA_TFLOPS, 4), traffic=None,
                    avg_us=round(ms * 1e3, 2), launches_timed=rec['n'])
    ach = rec['bytes'] / (ms * 1e-3) / 1e9
    return dict(kernel=name, bound='hbm', achieved=round(ach, 2), peak=PEAK_HBM_GBS, unit='GB/s',
                frac=round(ach / PEAK_HBM_GBS, 4), traffic=None, avg_us=round(ms * 1e3, 2),
                launches_timed=rec['n'])


def free_port():
    sk = socket.socket()
    sk.bind(('127.0.0.1', 0))
    port = sk.getsockname()[1]
    sk.close()
    return port


def count_gpus():
    """GPUs this process may use, WITHOUT touching HIP: the parent of an N-rank run must stay GPU-free (a process that
    has initialised the GPU may neither fork nor exec its ranks on this pool).  Source: the KFD topology in sysfs - a
    node with simd_count > 0 is a GPU - narrowed by HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES / CUDA_VISIBLE_DEVICES.
    Where sysfs has no KFD tree (a container without the driver's view) a throw-away CHILD asks torch; the parent
    itself never does."""
    root = '/sys/class/kfd/kfd/topology/nodes'
    n = None
    if os.path.isdir(root):
        n = 0
        for node in sorted(os.listdir(root)):
            try:
                with open(os.path.join(root, node, 'properties')) as f:
                    props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            except OSError:
                continue
            if int(props.get('simd_count', '0')) > 0:
                n += 1
    if n is None:
        r = subprocess.run([sys.executable, '-c', 'import torch; print(torch.cuda.device_count())'],
                           stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        try:
            n = int(r.stdout.strip().splitlines()[-1])
        except (ValueError, IndexError):
            n = 0
    for var in ('HIP_VISIBLE_DEVICES', 'ROCR_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        v = os.environ.get(var)
        if v is not None:
            listed = [x for x in v.split(',') if x.strip() != '']
            n = min(n, len(listed))
    return n


def launch_ranks(args, n_ranks=None, dry=False):
    """`bench.py --gpus N` (N > 1) outside a launcher: start N rank processes.  This parent NEVER initialises HIP
    (count_gpus reads sysfs; the product is not even imported), so there is no exec / fork of a GPU-initialised
    process: the ranks are fresh children of torch.distributed.run.  Their stdout is relayed: the ONE result line goes to
    this process's stdout, a non-zero exit code of any rank (the launcher's) becomes this process's."""
    n_ranks = n_ranks or args.gpus
    if not dry:
        have = count_gpus()
        if have < n_ranks:
            sys.stderr.write('bench.py: --gpus %d requested but this box exposes %d GPU(s); refusing to report a '
                             '%d-GPU line from fewer ranks\n' % (n_ranks, have, n_ranks))
            return 2
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_ranks),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    lines = [x for x in r.stdout.splitlines() if x.startswith('{') and '"metric"' in x]
    if r.returncode != 0 or not lines:
        sys.stderr.write('bench.py: the %d-rank run failed (launcher exit code %d, %d result line(s))\n'
                         % (n_ranks, r.returncode, len(lines)))
        if lines:
            # the headline was measured before a side measurement failed (HeadlineGuard wrote it): relay it, keep the code
            print(lines[-1], flush=True)
        else:
            sys.stderr.write(r.stdout[-2000:])
        return r.returncode or 1
    print(lines[-1], flush=True)
    return 0


def dry_rank(args):
    """One rank of `--dry-ranks N`: the launch path's plumbing on the CPU - rendezvous from the launcher's environment,
    a gloo group, one all-reduce, a barrier, rank 0's single JSON line - with neither a GPU nor the product."""
    import torch as torch_
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    if rank == args.dry_fail_rank:
        sys.stderr.write('bench.py: dry rank %d fails on request\n' % rank)
        os._exit(3)
    tok = torch_.tensor([float(rank + 1)])
    dist.all_reduce(tok)
    assert float(tok) == world * (world + 1) / 2, float(tok)
    dist.barrier()

    def compose(note=None):
        d = dict(metric='dry-run (launch path only: no GPU work)', value=float(tok), unit='token sum',
                 n_gpus=world, steps=0, warmup=0, ms_per_step=0.0, higher_is_better=True, scaling='weak',
                 vs_baseline=None, dtype='f32', data='none', config={'workload': 'launcher rehearsal'})
        if note is not None:
            d['extra'] = {'side_measurements_aborted': note}
        return json.dumps(d)
    if args.dry_fail_late_rank >= 0 or args.dry_hang:
        # the "side measurements" of the rehearsal: the headline (the token sum) exists; one rank dies, or rank 0 hangs
        if rank == args.dry_fail_late_rank:
            time.sleep(1.0)                       # (rank 0 is inside its wait by then)
            sys.stderr.write('bench.py: dry rank %d fails late on request\n' % rank)
            os._exit(3)
        if rank == 0:
            global _REAL_STDOUT
            _REAL_STDOUT = 1
            with HeadlineGuard(compose, args.extras_deadline) as guard:
                guard.job = 'dry side measurement'
                threading_event_wait_forever()
        else:
            threading_event_wait_forever()
    if rank == 0:
        print(compose(), flush=True)
    dist.destroy_process_group()
    return 0


def threading_event_wait_forever():
    """Stands in for a collective whose peer is gone (a C-level wait that no Python signal handler interrupts)."""
    import threading
    threading.Event().wait()


def main():
    args = parse()
    under_launcher = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ
    if args.dry_ranks:
        sys.exit(dry_rank(args) if under_launcher else launch_ranks(args, n_ranks=args.dry_ranks, dry=True))
    if under_launcher and int(os.environ['WORLD_SIZE']) != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started WORLD_SIZE=%s ranks\n'
                         % (args.gpus, os.environ['WORLD_SIZE']))
        sys.exit(2)
    if args.gpus > 1 and not under_launcher:
        sys.exit(launch_ranks(args))
    load_product()
    # stdout must carry exactly ONE JSON line: send everything libraries print (RCCL's version banner,
    # MIOpen notices ...) to stderr until the result is ready
    sys.stdout.flush()
    global _REAL_STDOUT
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    _REAL_STDOUT = saved_stdout          # (HeadlineGuard writes the line there if the side measurements never return)
    try:
        line, rc = run(args)
    finally:
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)
        _REAL_STDOUT = 1
        os.close(saved_stdout)
    if line is not None:
        print(line, flush=True)
    if rc:
        # rank 0 failed inside a side measurement under a group: the headline is out; no teardown (the peers sit in
        # collectives this rank will not join) - leave at once so that the launcher ends them
        sys.stderr.flush()
        os._exit(rc)


def run(args):
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local = int(os.environ.get('LOCAL_RANK', 0))
    under_launcher = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ
    if world > 1 or under_launcher:      # torchrun (also with one rank: exercises the RCCL path)
        import torch.distributed as dist
        torch.cuda.set_device(local)
        dist.init_process_group('nccl', device_id=torch.device('cuda', local))
        assert dist.get_world_size() == args.gpus, (dist.get_world_size(), args.gpus)
    dev = torch.device('cuda', local if (world > 1 or under_launcher) else 0)
    torch.cuda.set_device(dev)
    if args.scaling == 'strong':
        if args.batch % world:
            raise SystemExit('--scaling strong: --batch %d is not a multiple of %d ranks' % (args.batch, world))
        args.batch //= world
    B = args.batch
    ops.set_h3_mode(args.h3_mode)

    weights = synth.make_weights(V, synth.DEFAULT_SETTINGS, seed=0)
    cap = Captioner(synth.make_idx2word(V), synth.SENTIMENT_CATEGORIES, synth.DEFAULT_SETTINGS)
    cap.load_state_dict({k: torch.from_numpy(v) for k, v in weights.items()})
    cap.to(dev).eval()
    inputs, _ = device_inputs(B, 100 + rank, dev)

    def barrier():
        if world > 1 or under_launcher:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # The cyclic collector is off for warm-up AND timing (collected first): a full collection of a process with
    # torch imported costs 35-40 ms - inside the window it made the headline bimodal (173k vs 147k captions/s), and
    # right in front of it the GPU would start the timed region from idle clocks.
    with torch.no_grad(), no_gc():
        for w in range(args.warmup):
            # warm the per-kernel (event-timed) code paths too: the one-call step plan never touches them
            ops.TIMER.arm_step = None if args.no_kernel_timing else (w % 3) - 1
            cap(*inputs, T, 1, mode='rl')
        ops.TIMER.arm_step = None
        ops.TIMER.records.clear()
        barrier()
        t0 = time.perf_counter()
        for k in range(args.steps):
            # HIP events around every kernel of ONE decode step of this roll-out (-1 = prologue)
            ops.TIMER.arm_step = None if args.no_kernel_timing else (k % (T + 1)) - 1
            seq, lp, mk = cap(*inputs, T, 1, mode='rl')
        barrier()
        el = time.perf_counter() - t0
    ops.TIMER.arm_step = None
    if world > 1 or under_launcher:
        tt = torch.tensor([el], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        el = float(tt)
    # ---- the line's own part: everything the headline's measurements determine, BEFORE any side measurement runs
    total = world * B * args.steps
    summ = ops.TIMER.summary()
    # share of one roll-out: step kernels run T times, prologue kernels once
    entries = []
    for name, rec in summ.items():
        e = roofline_entry(name, rec)
        e['traffic'] = pmc_traffic(name, B)
        pc = pmc_counters(name, B)
        if pc:
            e['pmc'] = pc
        e['algorithmic'] = rec['flops'] if rec['flops'] > 0 else rec['bytes']
        e['phase'] = rec['phase']
        e['per_rollout_ms'] = round(rec['avg_ms'] * (1 if rec['phase'] == 'prologue' else T), 3)
        entries.append(e)
    # The armed step runs through the per-op host path (one FFI call and two event records per kernel), where the device
    # can go idle between launches: the event brackets then include those gaps and their sum exceeds the wall time of a
    # roll-out on the plan path (one FFI call per step) that `value` measures.  `per_rollout_ms` is therefore the
    # kernel's SHARE of the measured wall time (event time x wall / event sum when the sum is larger);
    # `per_rollout_ms_events` keeps the raw event figure, `avg_us` / `achieved` / `frac` are per launch from the events.
    wall_ms = el / args.steps * 1e3
    ev_sum = sum(e['per_rollout_ms'] for e in entries)
    scale = min(1.0, wall_ms / ev_sum) if ev_sum > 0 else 1.0
    for e in entries:
        e['per_rollout_ms_events'] = e['per_rollout_ms']
        e['per_rollout_ms'] = round(e['per_rollout_ms'] * scale, 3)
    entries.sort(key=lambda e: -e['per_rollout_ms'])
    # `roofline` is ALWAYS the attention scan (north_star names it; HBM-bound) and `roofline_mfma` the classifier (the one
    # contraction north_star puts on MFMA): the two tie at ~10 ms per roll-out, so "largest share" flipped from box to box
    scan = next((e for e in entries if e['kernel'].startswith('attn_scan')), None)
    mfma = next((e for e in entries if e['kernel'].startswith('vocab[')), None)
    rest = [e for e in entries if e is not scan and e is not mfma]
    extra = {}

    def compose(note=None):
        """The ONE result line: the headline + the side measurements finished so far.  `note`: why the side measurements
        were cut short (HeadlineGuard / a failure on this rank) - the headline was measured before they started."""
        ex = dict(extra)
        if note is not None:
            ex['side_measurements_aborted'] = note
        out = {
            'metric': 'captions/sec (greedy, 36-region feats, len-20)',
            'value': round(total / el, 1), 'unit': 'captions/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': round(el / args.steps * 1e3, 3),
            'higher_is_better': True, 'scaling': args.scaling, 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': 'greedy decode forward_rl(sample_max=1): B=%d captions/GPU/step, R=%d '
                                   'regions x 2048, V=%d, T=%d, sentiment-word attention + gate on, '
                                   'prologue included, random-init reference-architecture weights; the weight-only '
                                   'tables (relu(Emb) W_x^T [V,4H], sentiment-word tables 2 x [V,512]; 26 GFLOP) and '
                                   'the f16 weight planes are built by the first roll-out after a weight change, i.e. '
                                   'in warm-up, and reused by the timed ones (extra.table_build)' % (B, R, V, T),
                       'batch_per_gpu': B, 'global_batch': B * world,
                       'parallelism': 'dp%d (batch shard, no collective)' % world,
                       'ranks': torch.distributed.get_world_size() if dist_on() else 1,
                       'collective_backend': torch.distributed.get_backend() if dist_on() else None,
                       'gemm_engine': {1: 'split-f16 x3 MFMA for large forward GEMMs (fp32 in/out/accumulate), fp32 MFMA elsewhere',
                                       0: 'fp32 MFMA only (--h3-mode 0)', 2: 'split-f16 forced'}[args.h3_mode]},
            'roofline': scan if scan is not None else (entries[0] if entries else None),
            'roofline_mfma': mfma,
            'roofline_kernels': rest if scan is not None else entries[1:],
            'extra': ex,
        }
        if not args.no_cpu_baseline and world == 1 and note is None:
            out['cpu_baseline'] = cpu_baseline(weights, args.cpu_seconds)
        # LAST key: the driver's record keeps the tail of this line - the other half of BASELINE's metric (beam-5 p50) and
        # the training figures in one compact object
        g = lambda *ks: functools.reduce(lambda d, k: d.get(k) if isinstance(d, dict) else None, ks, ex)
        out['summary'] = {
            'greedy_captions_per_s': out['value'], 'scan_frac_of_hbm': (scan or {}).get('frac'),
            'classifier_frac_of_mfma': (mfma or {}).get('frac'),
            'beam5_p50_ms': g('beam5', 'per_image_p50_ms'), 'beam5_us_per_step': g('beam5', 'per_step_p50_us'),
            'beam5_full20_p50_ms': g('beam5', 'full_search_p50_ms'),
            'exact_fp32_captions_per_s': g('exact_fp32_engine', 'captions_per_s'),
            'xe128_ms': g('xe_train', 'ms_per_iter'), 'xe128_ss025_ms': g('xe_train_ss025', 'ms_per_iter'),
            'xe512_ms': g('xe_train_by_batch', '512'), 'xe1024_ms': g('xe_train_strong', 'ms_per_iter'),
            'xe1024_graph_ms': g('xe_train_strong', 'graph_ms_per_iter'),
            'exposed_allreduce_ms': (g('xe_train', 'exposed_allreduce_ms') if g('xe_train', 'exposed_allreduce_ms') is not None
                                     else g('xe_exchange_one_rank', 'exposed_allreduce_ms')),
            'exposed_allreduce_ranks': (world if g('xe_train', 'exposed_allreduce_ms') is not None
                                        else g('xe_exchange_one_rank', 'ranks')),
            'rl512_ms': g('rl_iteration', 'ms_per_iter'),
            'r196_greedy_captions_per_s': g('r196', 'greedy_B4096', 'captions_per_s'),
            'r196_scan_frac_of_hbm': g('r196', 'scan', '4096', 'frac_of_8tbs'),
            'r196_xe128_ms': g('r196', 'xe_train_B128', 'ms_per_iter'), 'r196_beam5_full20_p50_ms': g('r196', 'beam5', 'full_search_p50_ms'),
            'xe128_epoch_ms_per_iter': g('epoch_loops', 'xe128_resident_ms_per_iter'),
            'rl512_epoch_ms_per_iter': g('epoch_loops', 'rl512_resident_ms_per_iter'),
        }
        return json.dumps(out)

    def side_measurements(at):
        """`at(key)`: the job now running (HeadlineGuard's note)."""
        # secondary measurements (never part of `value`); failures are reported, not fatal
        if world == 1:
            # measurements on the headline's own weights first; the training benches (which update them) last
            for key, fn in (('exact_fp32_engine', lambda: bench_exact_fp32(cap, inputs, B)),
                            ('greedy_small_batches', lambda: bench_small_batches(cap, dev)),
                            ('batch_sweep', lambda: bench_batch_sweep(cap, dev)),
                            ('beam5', lambda: bench_beam(cap, inputs)),
                            ('scan_sweep', lambda: bench_scan_sweep(dev)),
                            ('table_build', lambda: bench_table_build(cap, inputs)),
                            ('rl_iteration', lambda: bench_rl(dev)),
                            ('rl_iteration_cold_sentiment_cache', lambda: bench_rl(dev, iters=3, cache_image_sentiments=False))):
                progress('extra: ' + key)
                at(key)
                try:
                    extra[key] = fn()
                except Exception as e:  # noqa: BLE001 - side measurements are reported, never fatal
                    extra[key] = {'error': repr(e)[:300]}
        # training curves of BASELINE configs[3] (XE) and [4] (RL); every rank runs them (they hold collectives).
        # weak: 128 captions + 80 seq2seq rows per GPU; strong: GLOBAL 1024 captions + 80 seq2seq rows over the ranks
        jobs = [('xe_train', lambda: bench_xe_train(cap, dev, rank, world))]
        if 1024 % world == 0 and 80 % world == 0:
            jobs.append(('xe_train_strong', lambda: bench_xe_train(cap, dev, rank, world, iters=4, B=1024 // world,
                                                                   s2s_rows=80 // world, curve='strong')))
        # the regime the reference trains in from epoch 5 on: scheduled sampling (train_xe.py:209-212, opts.py:35-38)
        jobs.append(('xe_train_ss025', lambda: bench_xe_train(cap, dev, rank, world, iters=8, ss_prob=0.25)))
        if world == 1:
            jobs.append(('xe_train_by_batch', lambda: {str(b): bench_xe_train(cap, dev, rank, world, iters=6, B=b)[
                'ms_per_iter'] for b in (512,)}))
            jobs.append(('r196', lambda: bench_r196(cap, dev)))
            jobs.append(('epoch_loops', lambda: bench_epoch_loops(dev)))
            if not dist_on():
                jobs.append(('xe_exchange_one_rank', bench_xe_exchange_one_rank))
        if dist_on():
            jobs.append(('grad_allreduce', lambda: bench_grad_allreduce(cap, dev, world)))
            if 512 % world == 0 and 80 % world == 0:
                jobs.append(('rl_iteration', lambda: bench_rl(dev, rank=rank, world=world)))
        run_jobs(jobs, extra, at)

    grouped = world > 1 or under_launcher
    if not args.no_extras:
        if grouped and rank == 0:
            # the headline must reach stdout even if a side measurement fails or hangs on ANY rank (HeadlineGuard)
            guard = HeadlineGuard(compose, args.extras_deadline)
            try:
                with guard:
                    side_measurements(lambda key: setattr(guard, 'job', key))
            except BaseException as e:  # noqa: BLE001 - FatalUnderGroup and whatever else ends this rank
                sys.stderr.write('bench.py: rank 0 left the side measurements in %r: %r\n' % (guard.job, e))
                return compose('rank 0 failed in %r: %s' % (guard.job, repr(e)[:300])), 1
        else:
            side_measurements(lambda key: None)
    if rank != 0:
        torch.distributed.destroy_process_group()
        return None, 0
    line = compose()
    if grouped:
        torch.distributed.destroy_process_group()
    return line, 0


if __name__ == '__main__':
    main()

"""Batches of frame pairs through the GPU back to back: the set-up of batch k+1 overlaps the optimisation of batch k.

One batch alone leaves the GPU idle twice: while the host lays out the tables between the count pass and the fill pass of the
set-up (``optim/batch_prepare.py``), and in the tail of the schedule, when the last few pairs iterate alone
(``PairBatch.run_scheduled``; ``profiles/r02_schedule_sweep.txt``).  Two host threads with one HIP stream each fill those gaps
with the other batch's work: a producer builds ``PairBatch`` objects one batch ahead on the set-up stream, the consumer runs
the schedule on the optimisation stream, an event orders "built" before "optimise", and a batch's arrays stay referenced until
its results have been read.  Frame pairs are independent problems, so nothing else is shared.

Measured (round 3, profiles/r03_stream_bench.txt: 640x480x64 pairs from raw frames, distinct frames per batch, a long-lived
PairStream): 384 pairs per batch 17.1 k pairs/s one batch at a time, 18.2 / 20.9 / 21.2 k with 1 / 2 / 3 schedules in flight
(``optimisers``); 128 pairs per batch 13.8 k -> 19.9 k.  In bench.py (the same frames every batch) 21.9 k -> 24-26 k.
Round 4 (profiles/r04_stream_bench.txt, r04_bench_n1.json; three rotating distinct input sets, reaper thread, level 0 on its stride-2
lattice): 21.2 k one batch at a time, 24.4 / 27.9 / 29.1 k with 1 / 2 / 3 schedules in flight in the tool; 33 k sustained in bench.py.
"""
from __future__ import annotations

import os
import queue
import sys
import threading
import time

import torch

from .pair_batch import FRAME_PAIR_POINT_STRIDE, FRAME_PAIR_SCHEDULE, PairBatch


class PairResult(tuple):
    """What ``PairStream.run`` yields per batch: unpacks as ``(poses, klds)``; ``.status`` = the (M,) int32 device tensor of SP_STATUS_*
    bits of the batch's scheduled run (include/sp_hip.h SpVerdict; 0 = converged, ``status & _lib.SP_STATUS_FAILED`` = do not trust the
    pair's pose and depths), ``.attempts`` = who was run a second time; both None when the schedule was run with ``verdict=False``."""

    def __new__(cls, poses, klds, status=None, attempts=None):
        self = super().__new__(cls, (poses, klds))
        self.status, self.attempts = status, attempts
        return self


class _SwitchInterval:
    """The interpreter's thread switch interval, lowered while any PairStream has host threads at work and put back when the last one is
    done -- or hands control to its caller (reference counted: nested / concurrent streams do not restore it under each other)."""
    _lock = threading.Lock()
    _count = 0
    _saved = None

    @classmethod
    def acquire(cls, value=2e-4):
        with cls._lock:
            if cls._count == 0:
                cls._saved = sys.getswitchinterval()
                sys.setswitchinterval(min(cls._saved, value))
            cls._count += 1

    @classmethod
    def release(cls):
        with cls._lock:
            cls._count -= 1
            if cls._count == 0:
                sys.setswitchinterval(cls._saved)


class PairStream:
    def __init__(self, levels=(0, 3), point_stride=FRAME_PAIR_POINT_STRIDE, schedule=None, device="cuda:0", depth=1, optimisers=1, **batch_kw):
        """``schedule``: keyword arguments of ``PairBatch.run_scheduled`` (default: FRAME_PAIR_SCHEDULE); ``depth``: how many
        built batches may wait for an optimiser (each holds its tables in device memory; peak residency is ``depth`` + 1 +
        ``optimisers`` batches: one being built, ``depth`` queued, one per optimiser); ``batch_kw``: passed to PairBatch.

        ``optimisers`` = K > 1: CONTINUOUS BATCHING at batch granularity.  K batches run their schedules at the same time, each on
        its own HIP stream and host thread.  A scheduled batch ends in a long tail -- the last ~10 % of its pairs iterate almost
        alone for a third of the rounds, at the latency of a two-launch iteration, with most of the chip idle
        (profiles/r02_schedule_sweep.txt) -- and the bulk phase of the next batch fills exactly that idle capacity: finished
        pairs' slots are, in effect, handed to fresh pairs without touching a single table or descriptor.  Pairs never interact
         and every batch keeps its own work lists and partial buffers, so each pair's result is bitwise
        the one it has when optimised alone.

        Meant to be long-lived: the caching allocator keeps one pool per stream, so a PairStream reuses its tables' memory
        from batch to batch, while a fresh one (fresh streams) pays for device allocations again."""
        self.levels, self.point_stride, self.batch_kw = levels, point_stride, batch_kw
        self.schedule = {k: v for k, v in (FRAME_PAIR_SCHEDULE if schedule is None else schedule).items() if k != "check_every"}
        self.device = torch.device(device)
        self.depth = depth
        # (equal priorities: giving the optimiser's chain of short launches a high-priority stream measured 9 % SLOWER at 384
        #  pairs per batch and collapsed at 64 -- tools/stream_bench.py)
        self.setup_stream = torch.cuda.Stream(self.device)
        self.optim_streams = [torch.cuda.Stream(self.device) for _ in range(max(1, int(optimisers)))]
        self.optim_stream = self.optim_streams[0]
        self.trace = [] if os.environ.get("SP_STREAM_TRACE") else None       # developer aid: (what, batch, host t0, host t1) per step
        self._reap = queue.Queue()
        self._reaper = None

    def _reaper_loop(self):
        """Frees finished batches off the optimiser threads: waits until the schedule's last launch is done, then drops the batch."""
        while True:
            item = self._reap.get()
            if item is None:
                return
            batch, done = item
            done.synchronize()
            del batch, item

    def _producer(self, inputs, out, ready_for_inputs, stop, n_consumers):
        def put(item):                       # never blocks for good: the consumers may have gone away
            while not stop.is_set():
                try:
                    out.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    pass
            return False

        try:
            torch.cuda.set_device(self.device)
            with torch.cuda.stream(self.setup_stream):
                self.setup_stream.wait_event(ready_for_inputs)
                for idx, item in enumerate(inputs):
                    if stop.is_set():
                        return
                    t0 = time.perf_counter()
                    batch = PairBatch(item["src_frames"], item["trg_images"], item["trg_Ks"], item["poses"], item["klds"], levels=self.levels,
                                      point_stride=self.point_stride, **self.batch_kw)
                    built = torch.cuda.Event()
                    built.record(self.setup_stream)
                    if self.trace is not None:
                        self.trace.append(("build", idx, t0, time.perf_counter()))
                    ok = put((idx, batch, built))
                    del batch                # the queue (then an optimiser) holds the only reference
                    if not ok:
                        return
            for _ in range(n_consumers):
                put(None)
        except BaseException as e:          # surfaces in the consumer
            for _ in range(n_consumers):
                put(e)

    def _optimiser(self, stream, built_q, results, caller, stop):
        """One optimiser: takes built batches, runs their schedules on its own stream, hands (index, poses, klds) over."""
        try:
            torch.cuda.set_device(self.device)
            while not stop.is_set():
                try:
                    got = built_q.get(timeout=0.1)
                except queue.Empty:
                    continue
                if got is None or isinstance(got, BaseException):
                    results.put(got)
                    return
                idx, batch, built = got
                with torch.cuda.stream(stream):
                    stream.wait_event(built)
                    t0 = time.perf_counter()
                    batch.run_scheduled(**self.schedule)
                    if self.trace is not None:
                        self.trace.append(("optimise", idx, t0, time.perf_counter()))
                    # (one copy of the flat log-depth array, handed out as per-pair views: a clone per pair is M launches)
                    poses, kld_flat = batch.poses().clone(), batch.kld.clone()
                    klds = [kld_flat[batch.n_off[m]: batch.n_off[m + 1]] for m in range(batch.M)]
                    # the results were allocated in this stream's pool and are consumed on the caller's stream: tell the
                    # allocator, so that a block the caller drops is not handed to a later batch's clone() while
                    # caller-stream work on it is still queued
                    poses.record_stream(caller)
                    kld_flat.record_stream(caller)
                    status = attempts = None
                    if batch.status is not None:                 # the run's verdict travels with its results
                        status, attempts = batch.status.clone(), batch.attempts.clone()
                        status.record_stream(caller); attempts.record_stream(caller)
                    done = torch.cuda.Event()
                    done.record(stream)
                # the batch (allocated on the set-up stream, used on this stream) is released only after the optimiser has finished with
                # it -- by the REAPER thread: waiting for `done` and tearing down a PairBatch (a few thousand tensor handles, ~3 ms of
                # interpreter time) here would keep this stream idle for that long after every batch
                results.put((idx, PairResult(poses, klds, status, attempts), None, done))
                self._reap.put((batch, done))
                del batch
        except BaseException as e:
            results.put(e)

    def optimise(self, batches, restore=False):
        """The schedules of ALREADY BUILT batches (PairBatch objects, their set-up complete), K at a time: every optimiser stream
        takes the next batch off the list as soon as it has finished one, so the long tail of a batch -- its last few pairs
        iterating almost alone -- overlaps the bulk of the next (continuous batching without any set-up work in the picture).
        ``restore``: reset each batch to its initial values first (benchmarks re-running the same batches).  Returns when every
        batch has finished (device synchronised)."""
        batches = list(batches)
        if len({id(b) for b in batches}) != len(batches):
            # the same PairBatch on two streams at once = restore_initial() / run_scheduled() racing on one set of poses, log-depths,
            # phases and partial records: run a repeated batch in a second call
            raise ValueError("PairStream.optimise: the same PairBatch appears twice in one call")
        nxt = [0]
        lock = threading.Lock()
        errors = []
        torch.cuda.synchronize(self.device)          # the batches may have been built / touched on other streams

        def worker(stream):
            try:
                torch.cuda.set_device(self.device)
                with torch.cuda.stream(stream):
                    while True:
                        with lock:
                            i = nxt[0]
                            nxt[0] += 1
                        if i >= len(batches):
                            break
                        if restore:
                            batches[i].restore_initial()
                        batches[i].run_scheduled(**self.schedule)
                stream.synchronize()
            except BaseException as e:               # noqa: BLE001
                errors.append(e)

        threads = [threading.Thread(target=worker, args=(st,), daemon=True) for st in self.optim_streams]
        _SwitchInterval.acquire()                                    # (see run())
        try:
            for t in threads:
                t.start()
            for t in threads:
                t.join()
        finally:
            _SwitchInterval.release()
        if errors:
            raise errors[0]

    def run(self, inputs):
        """inputs: iterable of dict(src_frames, trg_images, trg_Ks, poses, klds) -- the arguments of PairBatch, device resident.
        Yields a ``PairResult`` -- (poses (M,4,4), [klds]) with the run's per-pair ``.status`` / ``.attempts`` -- of every batch, in input
        order, as device tensors valid on the caller's current stream."""
        caller = torch.cuda.current_stream(self.device)
        ready = torch.cuda.Event()
        ready.record(caller)                 # the inputs may still be in flight on the caller's stream
        built_q = queue.Queue(maxsize=self.depth)
        results = queue.Queue()
        stop = threading.Event()
        K = len(self.optim_streams)
        workers = [threading.Thread(target=self._producer, args=(inputs, built_q, ready, stop, K), daemon=True)]
        workers += [threading.Thread(target=self._optimiser, args=(st, built_q, results, caller, stop), daemon=True) for st in self.optim_streams]
        # The producer is interpreter-bound for milliseconds at a time (a few thousand tensor handles per batch) and CPython hands
        # the GIL over only every sys.getswitchinterval() = 5 ms: an optimiser thread that has just come back from the native
        # schedule loop would wait that long before it can even take the next batch (measured: a 7 ms hole after every batch).
        # The lowered interval is process wide: it is reference counted over all streams at work and handed back whenever this
        # generator hands control to its caller (ADVICE r03 / VERDICT r04: the caller's own threads run at the interval they chose).
        _SwitchInterval.acquire()
        held = True
        reaper = threading.Thread(target=self._reaper_loop, daemon=True)
        reaper.start()
        for w in workers:
            w.start()
        try:
            pending, nxt, finished = {}, 0, 0
            while finished < K or pending:
                if nxt in pending:
                    res, _, done = pending.pop(nxt)
                    caller.wait_event(done)
                    nxt += 1
                    _SwitchInterval.release(); held = False
                    yield res
                    _SwitchInterval.acquire(); held = True
                    continue
                if finished == K:            # every optimiser has ended and the next index never came
                    raise RuntimeError("PairStream lost a batch")
                got = results.get()
                if got is None:
                    finished += 1
                elif isinstance(got, BaseException):
                    raise got
                else:
                    pending[got[0]] = got[1:]
        finally:                            # also when the caller abandons the generator early
            if held:
                _SwitchInterval.release()
            stop.set()
            for w in workers:
                w.join(timeout=60)
            self._reap.put(None)
            reaper.join(timeout=60)

"""Throughput pipeline of the hot path: stage 1 (equivariant net) of batch i+1 overlaps stage 2 (marker fit, 32 persistent
workgroups = 1/8 of the chip) of batch i on a second HIP stream, and the coordinate-only index ops of batch i+1 (the model's
index stream) start as soon as its points are resident.  Results are identical to the synchronous
`inference_demo.predict_smpl_batch`; only the schedule differs.  Scans are independent, so no state is shared between
in-flight batches besides the (read-only) weights and body-model tables."""
import torch

from . import ops
from .models.fit_SMPL import fit_smpl_device, fit_smpl_finalize, fit_smpl_stage_host


def limit_host_threads(n=None):
    """Cap torch's CPU intra-op thread pool for a process whose job is to feed the GPU.  Measured in round 6 (scratch/cgroup_probe.sh, thread_probe.sh on the
    pool's GPU boxes): the pipeline's host side is ONE enqueueing thread, but any small CPU tensor op wakes torch's OpenMP pool -- 128 threads on these
    hosts -- and the workers then spin; together they burned 12 - 16 CPUs, the container's CFS quota ran out in most 100 ms periods (cpu.stat: 58 of 81
    periods throttled) and the enqueueing thread was frozen with them for the rest of the period: the 80 - 100 ms "slow enqueue" stalls that rounds 4 - 5
    attributed to other tenants' load (640 - 860 instead of 900+ scans/s).  n: threads to keep (default: ETCH_HOST_THREADS or 4; 0 = leave the
    pool alone).  Returns the previous setting."""
    import os
    prev = torch.get_num_threads()
    if n is None:
        n = int(os.environ.get("ETCH_HOST_THREADS", "4"))
    if n > 0 and prev > n:
        torch.set_num_threads(n)
    return prev


class Ticket:
    keepalive = None

    def __init__(self, done, stage1, fit):
        self.done, self.stage1, self.fit = done, stage1, fit
        self.finalized = None           # host-side result, set when the ticket was retired early to free its pinned buffers


class HotPathPipeline:
    def __init__(self, args, model, gender="neutral", max_in_flight=3, stage1_streams=1, **fit_kwargs):
        self.args, self.model, self.gender, self.fit_kwargs = args, model, gender, fit_kwargs
        limit_host_threads()              # (see there: a spinning 128-thread CPU pool starves the one thread that feeds the GPU)
        # stage 1 of consecutive batches alternates over `stage1_streams` streams: with 2, the low-occupancy kernels of one
        # batch (deep Point-Transformer levels: a few dozen workgroups) fill behind the chip-wide kernels of the other
        from .utils.cu_streams import make_stream
        self.s1s = [make_stream("main") for _ in range(stage1_streams)]      # plain torch streams unless ETCH_CU_PARTITION asks for a CU partition (A/B switch)
        # (priorities: the two Point-Transformer nets' streams are high-priority queues; raising the index stream as well gained 0.5 % on the frozen-fit
        # workload but starved this stream's fit -- 32 - 96 long-running workgroups that each need a whole compute unit -- whenever the fit runs its full
        # schedule: 548 against 700 scans/s with well-posed markers; raising both: 660.  Both stay at normal priority; ETCH_*_STREAM_PRIORITY override.)
        import os
        self.s2 = make_stream("side", priority=int(os.environ.get("ETCH_STAGE2_STREAM_PRIORITY", "0")))
        self.max_in_flight = max_in_flight
        self.in_flight = []
        self._pinned = [None] * (max_in_flight + 1)     # ring of pinned host buffers, one set per batch in flight (+1 being read)
        self._n = 0
        self.host_times = [] if os.environ.get("ETCH_PIPE_TIMING") == "1" else None
        self._timeline = []              # ETCH_PIPE_TIMING=1: (stage-1 begin, stage-1 end, batch done) events per batch + the host time of the submit's end
        self.reserved_gib = 0.0          # what _reserve_allocator put into the streams' allocator pools (reported next to peak_hbm_gib)

    def __del__(self):
        try:        # batches still in flight keep their cross-stream tensors alive through their tickets: do not let them go before the GPU is done
            for t in self.in_flight:
                t.done.synchronize()
        except Exception:
            pass

    def _reserve_allocator(self):
        """Once, when the second batch is submitted (every stream of the path exists by then): give each stream's pool of the caching allocator a large
        free block.  The pools are per stream and grow on demand; with three batches in flight the interleaving of allocations and cross-stream frees
        keeps changing for dozens of steps, and every new segment is a hipMalloc on the enqueueing thread (measured: 13 new segments inside 16 timed
        steps, one of them a 78 ms stall -- two steps' worth of the host's lead).  ETCH_PIPE_RESERVE_GIB (default 2 per stream, 0 = off); the path's
        own peak is ~5 GiB of the GPU's 288."""
        import os
        gib = float(os.environ.get("ETCH_PIPE_RESERVE_GIB", "2"))
        if gib <= 0 or not torch.cuda.is_available():
            return
        streams = list(self.s1s) + [self.s2]
        side = getattr(self.model, "_side_stream", None)
        streams += [side] if side is not None else []
        streams += list(getattr(self.model, "_head_streams", None) or [])
        # an optimisation must never take the run down: the reservation is scaled to what the GPU has free -- at most a quarter of it over all
        # streams (a smaller GPU, or one shared by several ranks) -- and an allocation that fails anyway is skipped
        try:
            free_b, _ = torch.cuda.mem_get_info(self.args.device)
        except Exception:
            free_b = 0
        per = min(int(gib * 2 ** 30), int(free_b // (4 * max(1, len(streams)))))
        self.reserved_gib = 0.0
        if per < 2 ** 26:
            return
        for st in streams:
            try:
                with torch.cuda.stream(st):
                    blk = torch.empty(per, dtype=torch.uint8, device=self.args.device)
                    del blk
                self.reserved_gib += per / 2 ** 30
            except torch.cuda.OutOfMemoryError:
                break

    def submit(self, points):
        """Enqueue one batch (B,N,3) resident on the device; returns a Ticket.  Does not block on the GPU unless
        `max_in_flight` batches are already outstanding: the pinned host buffers are a ring of max_in_flight + 1 slots, so the
        oldest ticket is then retired first (host waits for it; its result is kept on the ticket for result())."""
        import time
        if self._n == 1:
            self._reserve_allocator()
        t0 = time.perf_counter()
        t1 = t0
        while len(self.in_flight) >= self.max_in_flight:
            oldest = self.in_flight.pop(0)
            oldest.done.synchronize()
            oldest.keepalive = None
            t1 = time.perf_counter()
            oldest.finalized = fit_smpl_finalize(oldest.fit)
        t2 = time.perf_counter()
        caller = torch.cuda.current_stream()
        s1 = self.s1s[self._n % len(self.s1s)]
        s1.wait_stream(caller)
        # Every tensor that crosses streams inside this batch (the input, the index tensors of the model's index stream, what the side streams' heads
        # read and return, what stage 2 reads) is kept alive on the ticket until the batch's LAST kernel has completed, instead of being
        # record_stream()-ed: no allocator events (which every later torch.empty of the enqueue thread would poll, models_pointcloud._cross_stream),
        # and the caller may still drop `points` right after submit().  The pipeline owns the ticket (self.in_flight) until it is retired.
        keep = [points]
        timing = self.host_times is not None
        ev_begin = None
        with torch.no_grad():
            with torch.cuda.stream(s1):
                if timing:       # ETCH_PIPE_TIMING=1: the stage-1 stream's own timeline (gap_report): where the batch's first / last stage-1 kernel sit
                    ev_begin = torch.cuda.Event(enable_timing=True)
                    ev_begin.record(s1)
                # the model's index stream waits for the producer of `points` (the caller's stream), not for s1's queue
                self.model.input_producer = caller
                self.model.defer_join = True          # the heads on side streams are joined on s2 below, not on s1
                self.model.keepalive = keep
                try:
                    results, _ = self.model(points, pred_items=["confidence", "direction", "magnitude"], direction_mode="standard_vector")
                finally:
                    self.model.input_producer, self.model.defer_join, self.model.keepalive = None, False, None
                joins = self.model.pending_join or []
                self.model.pending_join = None
                ready = torch.cuda.Event(enable_timing=timing)
                ready.record(s1)
            conf = results["confidences"]
            keep += [conf, results["part_labels"], results["direction"], results["magnitude"]]
            with torch.cuda.stream(self.s2):
                self.s2.wait_event(ready)
                for ev in joins:
                    self.s2.wait_event(ev)
                labels = ops.argmax_rows(results["part_labels"])
                inner = ops.inner_points(points.contiguous(), results["direction"], results["magnitude"], float(self.args.scale_magnitude))
                fit = fit_smpl_device(self.args, inner, labels, conf, self.gender, **self.fit_kwargs)
                # device -> host copies ride on s2 right behind this batch's fit (pinned, reused buffers): result() must
                # not enqueue anything on s2, where it would queue behind the NEXT batch's fit
                fit_smpl_stage_host(fit, self._pinned[self._n % len(self._pinned)])
                self._pinned[self._n % len(self._pinned)] = fit["host"]
                self._n += 1
                done = torch.cuda.Event(enable_timing=timing)
                done.record(self.s2)
        t = Ticket(done, results, fit)
        t.keepalive = keep
        if timing:
            self._timeline.append((ev_begin, ready, done, time.perf_counter()))
        self.in_flight.append(t)
        if self.host_times is not None:      # (wait for the oldest ticket, finalize it, enqueue this batch) in ms: diagnostics
            self.host_times.append(((t1 - t0) * 1e3, (t2 - t1) * 1e3, (time.perf_counter() - t2) * 1e3))
            ms = torch.cuda.memory_stats()
            cur = {k: ms.get(k, 0) for k in ("segment.large_pool.allocated", "segment.small_pool.allocated", "reserved_bytes.all.current", "num_alloc_retries")}
            prev = getattr(self, "_ms_prev", cur)
            if self.host_times[-1][2] > 25.0:
                import sys
                print("slow enqueue %.0f ms (batch %d): allocator deltas %s" % (self.host_times[-1][2], self._n, {k: cur[k] - prev[k] for k in cur}), file=sys.stderr)
            self._ms_prev = cur
        return t

    def gap_report(self, last=None):
        """ETCH_PIPE_TIMING=1, after a synchronize: the stage-1 stream's timeline from HIP events (no tracer attached) -- per batch the span of its stage 1
        (first to last kernel on the stage-1 stream), the idle gap between the end of batch i's stage 1 and the start of batch i+1's on that stream, and
        the lag of batch i's completion (stage 2 on its own stream) behind its stage 1.  -> dict of lists (ms)."""
        tl = self._timeline[-last:] if last else self._timeline
        if len(tl) < 2 or len(self.s1s) != 1:
            return None
        span = [a.elapsed_time(b) for a, b, _, _ in tl]
        gap = [tl[i][1].elapsed_time(tl[i + 1][0]) for i in range(len(tl) - 1)]
        lag = [b.elapsed_time(c) for _, b, c, _ in tl]
        period = [tl[i][1].elapsed_time(tl[i + 1][1]) for i in range(len(tl) - 1)]
        return dict(stage1_span_ms=span, stage1_gap_ms=gap, stage2_lag_ms=lag, period_ms=period)

    def result(self, ticket):
        """Wait for one batch and return the reference's fit_smpl tuple (meshes, markers, valid, smpl_info)."""
        if ticket.finalized is not None:
            return ticket.finalized
        ticket.done.synchronize()
        ticket.keepalive = None
        if ticket in self.in_flight:
            self.in_flight.remove(ticket)
        return fit_smpl_finalize(ticket.fit)

    def run(self, batches):
        """Process an iterable of batches with at most `max_in_flight` enqueued; yields results in order."""
        pending = []
        for pts in batches:
            pending.append(self.submit(pts))
            if len(pending) >= self.max_in_flight:
                yield self.result(pending.pop(0))
        while pending:
            yield self.result(pending.pop(0))


class GraphPipeline:
    """The same schedule -- `max_in_flight` batches on the device at once, stage 2 of a batch next to stage 1 of the next -- with the device work of a batch
    recorded ONCE per slot as a HIP graph (etch_amd.graph.GraphedHotPath: stage 1 with its side streams as branches, labels / inner points, marker fit,
    final LBS) and replayed with one host call.  Why: the eager pipeline needs ~430 launches = ~7 ms of Python per 33 ms step, and on a host whose cores
    are busy with other tenants those 7 ms become 20 - 60 ms -- the enqueueing thread, not the GPU, then sets the rate (640 - 860 instead of
    900+ scans/s).  A replay costs the host the runtime's own submission of the recorded nodes (C++, ~1 ms) and nothing else.  OPT-IN: on a quiet host
    the eager pipeline is much the faster of the two (946 - 986 against 707 - 716 scans/s, profiles/r06_schedule_ab.txt: replays on different streams do
    not overlap each other on this runtime), so this only pays on a host that cannot keep the eager pipeline fed.  Results are bit-identical (the same
    kernels on the same data: tests/test_gpu_pipeline.py).  One instance serves ONE batch shape (B, N): the graphs own static input / output buffers and private memory pools."""

    def __init__(self, args, model, B, N, gender="neutral", max_in_flight=3, want_trace=False, **fit_kwargs):
        from .graph import GraphedHotPath
        limit_host_threads()
        self.args, self.B, self.N = args, B, N
        fit_kwargs = dict(fit_kwargs, want_trace=want_trace)
        self.slots = [GraphedHotPath(args, model, B, N, gender, **fit_kwargs) for _ in range(max_in_flight)]
        self.streams = [torch.cuda.Stream() for _ in range(max_in_flight)]
        self.max_in_flight = max_in_flight
        self.busy = [None] * max_in_flight          # the ticket that currently owns slot k's static buffers
        self._pinned = [None] * max_in_flight
        self._n = 0
        self.reserved_gib = 0.0
        self.host_times = None

    def _retire(self, t):
        if t.finalized is None:
            t.done.synchronize()
            t.finalized = fit_smpl_finalize(t.fit)
            t.keepalive = None

    def submit(self, points):
        """Enqueue one batch (B,N,3) resident on the device; returns a Ticket.  The slot's previous batch is retired first (its results leave the static
        buffers as host copies / clones before the replay overwrites them)."""
        assert tuple(points.shape) == (self.B, self.N, 3), (tuple(points.shape), (self.B, self.N, 3))
        k = self._n % self.max_in_flight
        if self.busy[k] is not None:
            self._retire(self.busy[k])
        st = self.streams[k]
        st.wait_stream(torch.cuda.current_stream())
        with torch.no_grad(), torch.cuda.stream(st):
            dev = dict(self.slots[k].replay(points))
            # markers / valid are part of the result tuple and live in the slot's static buffers: cloned (stream-ordered behind the replay)
            dev["markers"], dev["valid"] = dev["markers"].clone(), dev["valid"].clone()
            fit_smpl_stage_host(dev, self._pinned[k])
            self._pinned[k] = dev["host"]
            done = torch.cuda.Event()
            done.record(st)
        t = Ticket(done, dev["results"], dev)
        t.keepalive = [points]
        self.busy[k] = t
        self._n += 1
        return t

    def result(self, ticket):
        self._retire(ticket)
        return ticket.finalized

    def run(self, batches):
        pending = []
        for pts in batches:
            pending.append(self.submit(pts))
            if len(pending) >= self.max_in_flight:
                yield self.result(pending.pop(0))
        while pending:
            yield self.result(pending.pop(0))


def choose_pipeline(args, model, sample_batch, gender="neutral", max_in_flight=3, schedule="eager", log=None, **kw):
    """-> (pipeline, report) for schedule "eager" (HotPathPipeline, the default) or "graph" (GraphPipeline; `sample_batch` gives its fixed shape).
    Measured (round 6, one MI355X, host load 11 - 20, bench.py --schedule ...): eager 946 / 986 scans/s, graph replay 707 / 716 -- three graphs replayed on
    three streams run at the SYNCHRONOUS schedule's rate (44 ms/step): on this runtime the replays of a 430-node graph do not overlap each other the way
    the eager streams do.  The graph pipeline only wins on a host so busy that the eager enqueue drops below ~700 scans/s (seen: 628 - 657 at load 50)
    and is therefore opt-in; an automatic choice by a warm-up calibration was tried and dropped (profiles/r06_schedule_ab.txt)."""
    B, N = int(sample_batch.shape[0]), int(sample_batch.shape[1])
    report = {"schedule_requested": schedule, "schedule": schedule}
    if schedule == "graph":
        gkw = {k: v for k, v in kw.items() if k != "stage1_streams"}
        return GraphPipeline(args, model, B, N, gender, max_in_flight=max_in_flight, **gkw), report
    assert schedule == "eager", schedule
    return HotPathPipeline(args, model, gender, max_in_flight=max_in_flight, **kw), report

"""Throughput pipeline of the hot path: stage 1 (equivariant net) of batch i+1 overlaps stage 2 (marker fit, 32 persistent
workgroups = 1/8 of the chip) of batch i on a second HIP stream, and the coordinate-only index ops of batch i+1 (the model's
index stream) start as soon as its points are resident.  Results are identical to the synchronous
`inference_demo.predict_smpl_batch`; only the schedule differs.  Scans are independent, so no state is shared between
in-flight batches besides the (read-only) weights and body-model tables."""
import torch

from . import ops
from .models.fit_SMPL import fit_smpl_device, fit_smpl_finalize, fit_smpl_stage_host


class Ticket:
    keepalive = None

    def __init__(self, done, stage1, fit):
        self.done, self.stage1, self.fit = done, stage1, fit
        self.finalized = None           # host-side result, set when the ticket was retired early to free its pinned buffers


class HotPathPipeline:
    def __init__(self, args, model, gender="neutral", max_in_flight=3, stage1_streams=1, **fit_kwargs):
        self.args, self.model, self.gender, self.fit_kwargs = args, model, gender, fit_kwargs
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
        with torch.no_grad():
            with torch.cuda.stream(s1):
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
                ready = torch.cuda.Event()
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
                done = torch.cuda.Event()
                done.record(self.s2)
        t = Ticket(done, results, fit)
        t.keepalive = keep
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

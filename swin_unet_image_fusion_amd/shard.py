"""Multi-GPU execution of the fusion forward: one process per GPU, batch sharding, one collective.

The reference has no distributed code.  In eval mode nothing couples samples
(LayerNorm per token, attention per window, BatchNorm running statistics), so image pairs shard
across ranks with replicated weights and the only exchange is an all-gather of the fused
(B/G,1,H,W) outputs — RCCL over xGMI when the backend is "nccl" (SURVEY.md §8e).  The same class
runs under gloo on CPU with any forward callable, which is how the N>1 path is tested without GPUs.
"""
from __future__ import annotations

from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def shard_bounds(batch: int, world_size: int, rank: int) -> Tuple[int, int, int]:
    """Contiguous split of `batch` pairs over `world_size` ranks -> (start, stop, padded_shard).
    Shards are ceil(batch/world) wide; trailing ranks may be short (or empty) and are padded for the
    collective, which needs equal contributions."""
    if batch < 0 or world_size <= 0 or not 0 <= rank < world_size:
        raise ValueError(f"bad shard request batch={batch} world={world_size} rank={rank}")
    per = -(-batch // world_size) if batch else 0
    start = min(rank * per, batch)
    stop = min(start + per, batch)
    return start, stop, per


class GatherHandle:
    """Result of ShardedFusion.step_async(): wait() returns the gathered tensor once the step (and its collective) is ordered before
    the caller's stream (GPU: a stream-level wait, the host does not block; gloo: a host wait)."""

    def __init__(self, tensor: torch.Tensor, work=None, event=None):
        self._tensor, self._work, self._event = tensor, work, event

    def wait(self) -> torch.Tensor:
        if self._work is not None:
            self._work.wait()
            self._work = None
        if self._event is not None:
            torch.cuda.current_stream(self._tensor.device).wait_event(self._event)
            self._event = None
        return self._tensor


def _spin_ms(streams, cycles: int) -> float:
    """Host-timed duration of one spin kernel (torch.cuda._sleep: a single thread) on each of `streams`, started together."""
    import time
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


def concurrent_streams(device, n: int, avoid=(), candidates: int = 12):
    """n torch streams whose kernels really run side by side, none sharing a hardware queue with a stream in `avoid`.

    HIP maps streams onto a few hardware queues (4 by default) in first-use order, not one each (measured on this image: the null
    stream and the 7th created stream share queue 1, the 3rd and 4th share queue 4, ...), and two streams on one hardware queue
    execute their kernels in order: lanes that land on one queue overlap nothing.  Which streams collide depends on what the process
    created before (RCCL's streams, other runners), so it is measured: two one-thread spin kernels on two streams take one spin time
    on different queues and two on the same queue.  Returns fewer than n streams if the device has no more distinct queues."""
    cycles = 200_000
    probe = torch.cuda.Stream(device=device)
    _spin_ms([probe], cycles)                                   # warm-up (first use creates the queue)
    one = min(_spin_ms([probe], cycles) for _ in range(3))
    if one < 0.2:                                               # aim at ~0.4 ms per spin: well above launch and sync noise
        cycles = int(cycles * 0.4 / max(one, 1e-3))
        one = min(_spin_ms([probe], cycles) for _ in range(3))

    def overlap(a, b) -> bool:
        # single and pair timed back to back (the spin counts shader clocks: its wall time moves with the clock), best of three
        _spin_ms([a, b], cycles)
        return min(_spin_ms([a, b], cycles) / max(_spin_ms([a], cycles), 1e-6) for _ in range(3)) < 1.5

    chosen = []
    pool = [probe] + [torch.cuda.Stream(device=device) for _ in range(candidates - 1)]
    for c in pool:
        if len(chosen) == n:
            break
        if all(overlap(c, o) for o in list(avoid) + chosen):
            chosen.append(c)
    return chosen


class _DeferredGather(GatherHandle):
    """Handle of an overlapped step whose all-gather has not been started yet (ShardedFusion.step_async): the runner starts it
    in_flight - 1 steps later, or wait() does — always oldest first, so every rank issues its collectives in the same order.
    Every handle must be waited for on every rank."""

    def __init__(self, runner, slot: int, lane_stream):
        super().__init__(runner._gathered[slot])
        self._runner, self._slot, self._lane_stream, self._issued = runner, slot, lane_stream, False

    def _start(self) -> None:
        if self._issued:
            return
        r, k = self._runner, self._slot
        with torch.cuda.stream(self._lane_stream):     # RCCL's stream waits for the lane's staging copy, not for the caller's stream
            work = r._all_gather(k)
        r._slot_work[k] = self._work = work
        self._issued = True

    def wait(self) -> torch.Tensor:
        self._runner._start_through(self)
        return super().wait()


class _Lane:
    """One captured forward: the hipGraph, the runner-owned static input / output buffers it was captured on, the key it was captured
    for, its capture stream (also the stream it replays on when steps overlap) and the library workspace it points into."""
    __slots__ = ("graph", "static", "key", "stream", "ws_ref")

    def __init__(self):
        self.graph = self.static = self.key = self.stream = self.ws_ref = None


class ShardedFusion:
    """step(ir, vis): run this rank's pairs and return the fused output of ALL ranks, rank-major.

    step_async(ir, vis) returns a GatherHandle instead: the all-gather runs on the process group's own stream (RCCL over
    xGMI) while the caller enqueues the NEXT step's forward, and only handle.wait() orders it before the caller's stream.
    The local output is first copied into a staging buffer (the captured graph rewrites its output buffer at its next replay)
    and gathered into a result buffer of a small ring, so a handle stays valid until `in_flight` + 1 steps after it.

    forward_fn defaults to `model(ir, vis)` (the HIP path).  With `use_graph` the forward is captured
    into a hipGraph (torch.cuda.CUDAGraph on ROCm) and replayed: ~100 kernel launches per forward
    collapse into one graph launch.  The collective stays outside the graph.

    in_flight > 1 (graph mode on a GPU only): that many captured copies of the forward ("lanes"), each with its own static buffers,
    workspace and stream; consecutive step_async() calls go to consecutive lanes and run CONCURRENTLY until their handles are
    waited for.  One forward is a chain of ~90 dependent launches whose deep levels occupy a fraction of the chip for their latency
    (DESIGN 5): a second step in flight runs its wide level-0 launches under them.  Steps are independent batches, so results do
    not change; a lane's output buffer is rewritten `in_flight` steps later.  step() / local_forward() wait at once and so stay
    serial.

    A captured graph bakes in raw addresses: the model's weight arena and packed images, the library
    workspace, the arithmetic mode and the static input / output buffers.  It is therefore keyed on
    (input shape, model.graph_key()) and re-captured whenever either changes — after load_state_dict(),
    refresh_weights(), .to() or `model.precision = ...` the next step runs the new weights (a017:50-54: load,
    then infer).  The static input buffers belong to the runner (callers' tensors are copied in, never
    adopted), and the tensor returned by local_forward()/step() at world_size 1 is the runner's static
    output buffer: it is overwritten by a later step — clone it to keep it."""

    def __init__(self, model=None, world_size: int = 1, rank: int = 0, use_graph: bool = False,
                 forward_fn: Optional[Callable] = None, group=None, force_collective: bool = False, in_flight: int = 1):
        # force_collective: issue the all-gather even at world_size 1 (a one-rank RCCL group exercises the GPU branch on a
        # single-GPU box: tests/test_gpu_parity.py)
        if in_flight < 1:
            raise ValueError(f"in_flight must be >= 1, got {in_flight}")
        self.model, self.world_size, self.rank, self.group = model, world_size, rank, group
        self.force_collective = force_collective
        self.forward_fn = forward_fn or (lambda a, b: model(a, b))
        self.use_graph = use_graph
        self.in_flight = in_flight
        # several forwards in flight: batches that fill the chip take the kernel shapes with the least CU-time per forward
        # (model.schedule = "throughput", chosen at the first capture from the shard's size; the model's eager forwards follow, so replay
        # and eager stay bit-identical).  A schedule the caller set by hand is left alone.
        self._auto_schedule = in_flight > 1 and use_graph and getattr(model, "schedule", None) == "latency"
        self.graph_active = False
        self._lanes = [_Lane() for _ in range(in_flight)]
        self._lane_streams_ready = False
        self._turn = 0
        nslots = in_flight + 1   # step i+1 .. i+in_flight may run while step i's result is still being gathered / read
        self._gathered = [None] * nslots
        self._stage = [None] * nslots
        self._slot_work = [None] * nslots
        self._slot = 0
        self._unissued = []      # deferred collectives of overlapped steps, oldest first
        self.captures = 0

    # (the first lane under its old names: tests and tools look at them)
    _graph = property(lambda self: self._lanes[0].graph)
    _static = property(lambda self: self._lanes[0].static)
    _ws_ref = property(lambda self: self._lanes[0].ws_ref)
    _cap_stream = property(lambda self: self._lanes[0].stream)

    # -- forward of the local shard --------------------------------------------------------------
    def _model_key(self):
        return self.model.graph_key() if hasattr(self.model, "graph_key") else None

    def _capture(self, lane: _Lane, ir, vis):
        ir, vis = ir.clone(), vis.clone()             # runner-owned static buffers
        lane.graph = lane.static = lane.ws_ref = None      # drop the old graph before its buffers
        self.forward_fn(ir, vis)                      # warm-up: sizes the workspace, builds the arena, first-forward check
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        if lane.stream is None:                       # one capture stream for the lane's lifetime: the library workspace is keyed by stream
            lane.stream = torch.cuda.Stream(device=ir.device)
        s = lane.stream
        s.wait_stream(torch.cuda.current_stream(ir.device))
        with torch.cuda.stream(s):
            self.forward_fn(ir, vis)                  # warm-up on the capture stream (its own workspace)
            torch.cuda.synchronize()
            # thread_local: other threads of the process (the RCCL watchdog polls events) may call the runtime during the capture
            with torch.cuda.graph(g, stream=s, capture_error_mode="thread_local"):
                out = self.forward_fn(ir, vis)
        torch.cuda.current_stream(ir.device).wait_stream(s)
        # The graph holds the raw address of the library workspace registered for (device, capture-stream handle).  That
        # registry replaces (and so frees) a buffer when a later call on the same handle needs more bytes, and torch hands
        # stream handles out of a 32-entry pool — another runner or model with a larger shape may land on this handle.
        # Holding the tensor keeps the captured address valid for as long as this graph exists.
        try:
            from .modules import _workspace_tensor
            lane.ws_ref = _workspace_tensor(ir.device, s.cuda_stream)
        except ImportError:       # forward_fn-only runners (tests on CPU) never get here: capture needs CUDA tensors
            lane.ws_ref = None
        lane.graph, lane.static, self.graph_active = g, (ir, vis, out), True
        lane.key = (tuple(ir.shape), self._model_key())   # after the capture: the arena / packed images exist now
        self.captures += 1

    def _pick_lane_streams(self, device):
        """Overlapped lanes need streams on distinct hardware queues (concurrent_streams); with fewer distinct queues than lanes the
        runner keeps as many lanes as it found queues for."""
        # A lane may share its hardware queue with the caller's stream: what the caller puts there are waits for the OLDEST step in
        # flight, and everything queued in front of such a wait is work that finishes after that step anyway (measured: four lanes
        # on the four queues 9 025-9 041 pairs/s, three lanes that leave the caller's queue alone 8 635, same box).
        got = concurrent_streams(device, self.in_flight)
        if not got:
            got = [torch.cuda.Stream(device=device)]
        if len(got) < self.in_flight:
            self._lanes, self.in_flight = self._lanes[:len(got)], len(got)
            self._turn = 0
        for lane, s in zip(self._lanes, got):
            lane.stream = s
        self._lane_streams_ready = True

    # pixels per shard from which the throughput schedule pays (B=16 256x256: +4 %; B=1 256x256 is 10 % faster in the latency schedule
    # even with lanes: its launches leave most CUs empty either way; B=1 640x512 is even)
    _THROUGHPUT_FROM_PIXELS = 400_000

    def _ready_lane(self, lane: _Lane, ir, vis) -> _Lane:
        if self.in_flight > 1 and not self._lane_streams_ready:
            self._pick_lane_streams(ir.device)
            lane = self._lanes[self._turn]
        if self._auto_schedule and self.in_flight > 1:
            self._auto_schedule = False
            if ir.shape[0] * ir.shape[-2] * ir.shape[-1] >= self._THROUGHPUT_FROM_PIXELS:
                self.model.schedule = "throughput"
        if lane.static is None or lane.key != (tuple(ir.shape), self._model_key()):
            self._capture(lane, ir, vis)
        return lane

    def _issue(self, ir, vis) -> _Lane:
        """Next lane in turn: inputs copied in and the graph replayed on the LANE's stream, after everything the caller's stream holds
        so far (input producers; consumers of this lane's previous output).  Nothing waits for it here."""
        lane = self._ready_lane(self._lanes[self._turn], ir, vis)
        self._turn = (self._turn + 1) % self.in_flight
        s_ir, s_vis, _ = lane.static
        lane.stream.wait_stream(torch.cuda.current_stream(ir.device))
        with torch.cuda.stream(lane.stream):
            s_ir.copy_(ir)
            s_vis.copy_(vis)
            lane.graph.replay()
        ir.record_stream(lane.stream)
        vis.record_stream(lane.stream)
        return lane

    def _overlapped(self, ir) -> bool:
        return self.in_flight > 1 and self.use_graph and ir.is_cuda

    def local_forward(self, ir, vis):
        if not (self.use_graph and ir.is_cuda):
            return self.forward_fn(ir, vis)
        if self._overlapped(ir):
            lane = self._issue(ir, vis)
            torch.cuda.current_stream(ir.device).wait_stream(lane.stream)
            return lane.static[2]
        lane = self._ready_lane(self._lanes[0], ir, vis)
        s_ir, s_vis, s_out = lane.static
        s_ir.copy_(ir)
        s_vis.copy_(vis)
        lane.graph.replay()
        return s_out

    # -- collective ------------------------------------------------------------------------------
    def _next_slot(self, like: torch.Tensor) -> int:
        self._slot = (self._slot + 1) % len(self._stage)
        k = self._slot
        shape = (self.world_size * like.shape[0],) + tuple(like.shape[1:])
        if self._gathered[k] is None or self._gathered[k].shape != shape or self._gathered[k].device != like.device:
            self._gathered[k] = torch.empty(shape, dtype=like.dtype, device=like.device)
            self._stage[k] = torch.empty(tuple(like.shape), dtype=like.dtype, device=like.device)
        if self._slot_work[k] is not None:   # the slot's previous collective reads the staging buffer: order this stream behind it
            self._slot_work[k].wait()
        return k

    def _all_gather(self, k: int):
        """The collective of ring slot k: RCCL's all_gather_into_tensor; the list form under gloo (CPU tensors, and GPU tensors in the
        two-rank test that shares one GPU)."""
        if self._stage[k].is_cuda and dist.get_backend(self.group) == "nccl":
            return dist.all_gather_into_tensor(self._gathered[k], self._stage[k], group=self.group, async_op=True)
        return dist.all_gather(list(self._gathered[k].chunk(self.world_size, dim=0)), self._stage[k], group=self.group, async_op=True)

    def gather_async(self, local_out: torch.Tensor) -> GatherHandle:
        if self.world_size == 1 and not self.force_collective:
            return GatherHandle(local_out)
        k = self._next_slot(local_out)
        self._stage[k].copy_(local_out)   # the collective reads a buffer nothing rewrites while it is in flight
        work = self._all_gather(k)
        self._slot_work[k] = work
        return GatherHandle(self._gathered[k], work)

    def gather(self, local_out: torch.Tensor) -> torch.Tensor:
        return self.gather_async(local_out).wait()

    def _start_through(self, handle) -> None:
        """Start the deferred collectives up to and including `handle`, oldest first (every rank starts them in the same order)."""
        while self._unissued and not handle._issued:
            self._unissued.pop(0)._start()

    def step_async(self, ir_shard: torch.Tensor, vis_shard: torch.Tensor) -> GatherHandle:
        if not self._overlapped(ir_shard):
            return self.gather_async(self.local_forward(ir_shard, vis_shard))
        lane = self._issue(ir_shard, vis_shard)
        if self.world_size == 1 and not self.force_collective:
            return GatherHandle(lane.static[2], event=lane.stream.record_event())
        with torch.cuda.stream(lane.stream):      # the staging copy is ordered behind the lane's graph
            k = self._next_slot(lane.static[2])
            self._stage[k].copy_(lane.static[2])
        # The collective itself starts in_flight - 1 steps LATER.  RCCL's stream shares one of the 4 hardware queues with a lane (or
        # the caller); started now, its wait for THIS step would sit in that queue in front of the other lane's next graph and hold
        # it until this step is done - the lanes would run one after the other (measured: 3 lanes + immediate gather = 6 670 pairs/s,
        # the one-chain rate).  Started after the next in_flight - 1 steps are enqueued, everything queued in front of it is work that
        # runs before this step completes anyway.
        handle = _DeferredGather(self, k, lane.stream)
        self._unissued.append(handle)
        while len(self._unissued) > self.in_flight - 1:
            self._unissued.pop(0)._start()
        return handle

    def step(self, ir_shard: torch.Tensor, vis_shard: torch.Tensor) -> torch.Tensor:
        return self.step_async(ir_shard, vis_shard).wait()

    def fuse_global(self, ir: torch.Tensor, vis: torch.Tensor) -> torch.Tensor:
        """Every rank passes the same global batch; returns the full fused batch on every rank."""
        b = ir.shape[0]
        start, stop, per = shard_bounds(b, self.world_size, self.rank)
        if per == 0:
            return ir.new_empty((0, 1) + tuple(ir.shape[2:]))
        li, lv = ir[start:stop], vis[start:stop]
        if stop - start < per:    # pad short shards by repeating the last available pair; trimmed below
            fill = per - (stop - start)
            src_i = li[-1:] if stop > start else ir[-1:]
            src_v = lv[-1:] if stop > start else vis[-1:]
            li = torch.cat([li, src_i.expand(fill, *src_i.shape[1:])])
            lv = torch.cat([lv, src_v.expand(fill, *src_v.shape[1:])])
        out = self.step(li.contiguous(), lv.contiguous())
        if self.world_size == 1:
            return out[:b]
        keep = [out[r * per: r * per + max(0, min(per, b - r * per))] for r in range(self.world_size)]
        return torch.cat(keep)

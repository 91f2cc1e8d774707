"""Training-step driver: flat parameter/gradient buffers, fused AdamW, data parallel.

Replaces, for the hot path only, what the reference delegates to Lightning
(SURVEY.md §2.2-2.3): ``DistributedDataParallel``'s bucketed gradient all-reduce
(one process per GPU, RCCL over xGMI through ``torch.distributed`` backend
"nccl") and ``torch.optim.AdamW(lr, betas=(0.9, 0.95))`` (models/module.py:293-304).

MI355X-first choices:
  * all parameters live in ONE flat fp32 buffer and all gradients in another, so
    the optimizer is a single HBM-bound kernel (``nlam_adamw_step``) instead of a
    multi-tensor launch per ~100 small tensors, and a gradient bucket is a
    contiguous slice: the all-reduce needs no flatten / unflatten copies;
  * buckets are laid out in reverse registration order (= the order backward
    produces them).  Eager step: a bucket's all-reduce is launched as soon as its
    last gradient has landed -- the fused-MLP backward reports each finished
    accumulation (``ops.GRAD_LISTENER``; parameters that go through autograd's own
    accumulation report through post-accumulate hooks) -- so RCCL overlaps the
    rest of backward.  HIP-graph step (the default of bench.py): the collective
    is NOT overlapped: one all-reduce of the whole flat buffer follows each
    replay (0.86 MB at cfg2, 20.6 MB at cfg3; RCCL calls stay outside the
    capture so a communicator fault can never poison the graph).  The 1/world
    scaling is folded into the AdamW kernel;
  * xGMI is point-to-point: a ring all-reduce is per-link bound, so buckets are
    large (default 32 MiB) -- at cfg2/cfg3 sizes (0.86 / 20.6 MB) that is one
    collective per step.
"""
from __future__ import annotations

import contextlib

import torch
import torch.distributed as dist
from torch import nn


class FlatParams:
    """Re-homes every trainable parameter of ``module`` into one flat buffer
    (and its gradient into a second one) without changing names or shapes."""

    def __init__(self, module: nn.Module):
        self.params = [p for p in module.parameters() if p.requires_grad]
        if not self.params:
            raise ValueError("module has no trainable parameters")
        dev, dt = self.params[0].device, self.params[0].dtype
        # reverse registration order ~ the order in which backward finishes them
        self.order = list(reversed(range(len(self.params))))
        self.offsets = {}
        off = 0
        for i in self.order:
            self.offsets[i] = off
            off += (self.params[i].numel() + 3) // 4 * 4   # every view starts on a 16-byte boundary (vector loads / stores)
        self.numel = off
        self.flat = torch.zeros(off, device=dev, dtype=dt)
        self.grad = torch.zeros(off, device=dev, dtype=dt)
        for i, p in enumerate(self.params):
            o, n = self.offsets[i], p.numel()
            self.flat[o : o + n].copy_(p.data.reshape(-1))
            p.data = self.flat[o : o + n].view(p.shape)
            p.grad = self.grad[o : o + n].view(p.shape)

    def zero_grad(self):
        self.grad.zero_()
        for i, p in enumerate(self.params):  # autograd may have replaced .grad; re-attach the views
            o, n = self.offsets[i], p.numel()
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o : o + n].view(p.shape)


class GradBuckets:
    """Contiguous slices of the flat gradient buffer, all-reduced as they complete."""

    def __init__(self, fp: FlatParams, bucket_bytes: int = 32 << 20, group=None, comm_dtype=None):
        self.fp, self.group = fp, group
        # gradient exchange precision: None = the fp32 buffer in place; torch.bfloat16 halves the bytes on the xGMI links
        # (the "bf16 grads option" of SURVEY.md section 8(f)3): each bucket is cast, summed in bf16 and written back
        self.comm_dtype = comm_dtype
        self._comm_tmp = []
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.bounds = []  # (start, end, [param indices])
        cur_start, cur_members, cur_bytes = 0, [], 0
        for i in fp.order:
            n = fp.params[i].numel()
            cur_members.append(i)
            cur_bytes += 4 * n
            if cur_bytes >= bucket_bytes:
                end = fp.offsets[i] + n
                self.bounds.append((cur_start, end, cur_members))
                cur_start, cur_members, cur_bytes = end, [], 0
        if cur_members:
            self.bounds.append((cur_start, fp.numel, cur_members))
        self.bucket_of = {i: b for b, (_, _, mem) in enumerate(self.bounds) for i in mem}
        self.index_of = {id(p): i for i, p in enumerate(fp.params)}
        self.pending = [0] * len(self.bounds)
        self.uses = {}        # parameter index -> fused-MLP forwards that used it and have not yet back-propagated
        self.fired = set()    # parameters whose post-accumulate hook has fired while fused-MLP uses were still outstanding
        self.done = set()
        self.handles = []
        self.launched = []    # bucket indices in launch order (tests read it)
        self.enabled = True   # False while a HIP graph owns the step (collectives run after the replay)
        self.join_streams = lambda: None   # set by the trainer: make the launching stream wait for every producer stream
        # Completion signal of a parameter's gradient.  With hooks registered (world > 1) it is ALWAYS autograd's
        # post-accumulate hook: the AccumulateGrad node of a parameter runs once per backward, after every backward node that
        # feeds it has been issued -- including a fused MLP that accumulated the gradient itself and returned None for it
        # (the hook fires for an undefined gradient too) and including any plain torch op sharing the parameter.  The
        # fused ops' own note_done reports are then ignored: a parameter fed by both mechanisms would otherwise be marked
        # complete by whichever finishes first, and the late contribution would land after the bucket's collective started.
        # One case needs BOTH signals: a parameter back-propagated in the main graph and again in a nested backward
        # (ops.early_backward_leaf) gets an AccumulateGrad firing per backward call, and the first one may precede the nested
        # contribution -- so a hook that fires while fused-MLP forwards of the parameter are still un-back-propagated
        # (``uses`` > 0) only marks it, and the last note_done completes it.
        self.hooks_armed = False
        if self.world > 1:
            for i, p in enumerate(fp.params):
                p.register_post_accumulate_grad_hook(self._make_hook(i))
            self.hooks_armed = True

    def _reduce_slice(self, s, e):
        """Launch the all-reduce of grad[s:e]; in a reduced communication dtype the sum happens on a cast copy that
        ``finish_step`` writes back."""
        view = self.fp.grad[s:e]
        if self.comm_dtype is None or self.comm_dtype == view.dtype:
            self.handles.append(dist.all_reduce(view, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
            return
        low = view.to(self.comm_dtype)
        self.handles.append(dist.all_reduce(low, op=dist.ReduceOp.SUM, group=self.group, async_op=True))
        self._comm_tmp.append((view, low))

    def all_reduce_whole(self):
        """One collective over the whole flat gradient buffer (the HIP-graph step)."""
        self._reduce_slice(0, self.fp.numel)
        self._wait_all()

    def _wait_all(self):
        for h in self.handles:
            h.wait()
        self.handles = []
        for view, low in self._comm_tmp:
            # `low` was allocated on whichever stream was current when the bucket completed (a weight-gradient side stream in
            # the eager direct-gradient path); this copy runs on the current stream: tell the caching allocator, or the
            # block could be handed back to the side stream while the copy is still pending
            if low.is_cuda:
                low.record_stream(torch.cuda.current_stream())
            view.copy_(low)
        self._comm_tmp = []

    def _param_done(self, i):
        if i in self.done:
            return
        self.done.add(i)
        b = self.bucket_of[i]
        self.pending[b] -= 1
        if self.pending[b] == 0:
            s, e, _ = self.bounds[b]
            self.join_streams()   # the bucket's gradients were written from several streams
            self.launched.append(b)
            self._reduce_slice(s, e)

    def _make_hook(self, i):
        def hook(_param):   # autograd accumulated this parameter's gradient (complete unless fused uses are outstanding)
            if self.enabled:
                if self.uses.get(i, 0) > 0:
                    self.fired.add(i)
                else:
                    self._param_done(i)

        return hook

    # ---- ops.GRAD_LISTENER protocol: parameters whose gradients the fused-MLP backward accumulates itself ----
    def note_use(self, params):
        if not self.enabled or self.world == 1:
            return
        for p in params:
            i = self.index_of.get(id(p))
            if i is not None:
                self.uses[i] = self.uses.get(i, 0) + 1

    chain_recorder = None   # a set while a _SegmentedStep records its chain: ids of parameters whose .grad a CHAIN launch wrote

    def note_done(self, params):
        if self.chain_recorder is not None:
            self.chain_recorder.update(id(p) for p in params)
        if not self.enabled or self.world == 1:
            return
        for p in params:
            i = self.index_of.get(id(p))
            if i is None:
                continue
            left = self.uses.get(i, 1) - 1
            self.uses[i] = left
            if left <= 0 and (not self.hooks_armed or i in self.fired):   # the last contribution is in: the gradient is complete
                self._param_done(i)

    def begin_step(self):
        self.pending = [len(mem) for (_, _, mem) in self.bounds]
        self.uses, self.done, self.fired = {}, set(), set()
        self.handles, self.launched = [], []

    def finish_step(self):
        """Wait for the collectives; buckets whose hooks never fired (parameters
        without gradient this step) are reduced here so every rank issues the same
        sequence of collectives."""
        if self.world == 1:
            return
        for b, left in enumerate(self.pending):
            if left > 0:
                s, e, _ = self.bounds[b]
                self.launched.append(b)
                self._reduce_slice(s, e)
                self.pending[b] = 0
        self._wait_all()


class _SegmentedStep:
    """The captured training step as a CHAIN of HIP graphs on one stream plus weight-gradient graphs on side
    streams (DESIGN.md finding 39).

    Why not one graph with forks: the HIP-graph executor of ROCm 7.2 gives the k-th child of a node hardware queue
    ``(queue + k) mod 4`` (finding 38).  With the weight gradients of a fused MLP forked behind its data-gradient kernel the
    backward chain changes queue at every MLP (a 12 us signal gap instead of a 5 us back-to-back launch: 1 275 us of backward
    chain at cfg2 against 908 us on one queue), and every fork order that keeps the chain on one queue starts the side work
    late and in reverse.  Here the executor never sees a fork:

    * the chain -- zero-grad, weight packing, forward, loss, every data-gradient kernel -- is recorded as ``S`` LINEAR graphs
      (segments), replayed back to back on ONE stream (normal priority: on a high-priority stream -- its own hardware queue,
      ROCclr keeps a queue pool per priority -- the step took 4-5 ms instead of 1.9: the priority starves the side queues);
    * a fork point (``ops.OVERLAP.run``) records nothing: its closure is kept, and after ``K`` fork points (or behind a
      dead-end MLP, whose data-gradient kernel is side work too) the chain segment is closed;
    * after the chain has been recorded the closures of segment ``i`` are recorded into one linear graph per side stream
      ``k`` (the side stream is a function of the MLP, so a rollout's read-modify-write accumulations stay ordered); at
      replay stream ``k`` waits for the event recorded behind chain segment ``i`` and launches its graph: weight gradients
      of segment ``i`` run beside chain segment ``i + 1``, in order, from ordinary stream / event semantics;
    * the optimizer is a last graph behind the join (``1 / world`` folded in: at world > 1 the gradient all-reduce runs
      between the join and it, bucket by bucket as the segments that complete a bucket finish).

    Chain graphs share one private memory pool (they replay in recording order on one stream); the side graphs have
    their own: they run concurrently with later chain segments, so a block the chain has released must not be theirs.
    Everything a side closure reads is referenced until the last side graph is recorded (``ops.OVERLAP.keep``).
    """

    def __init__(self, trainer, forks_per_segment: int):
        import weakref

        self.tr = weakref.proxy(trainer)   # (the trainer owns this object: no reference cycle, its graphs' pools go when it goes)
        self.K = max(1, int(forks_per_segment))
        self.chain = []            # CUDAGraph per segment
        self.jobs = [[]]           # per segment: [(side stream index, closure)]
        self.side = []             # per segment: {stream index: CUDAGraph}
        self.events = []           # per segment: event recorded behind the chain segment (None: no side work)
        self.done_events = []      # per segment: {stream index: event recorded behind the side graph}
        self.tail = None           # the optimizer's graph (world == 1 or a capturable optimizer), replayed behind the join
        self.cur = None
        self.nforks = 0
        self.pool = torch.cuda.graph_pool_handle()
        self.side_pool = torch.cuda.graph_pool_handle()
        self.seg_params = []       # per segment: ids of the parameters whose gradients its side work completed
        self.chain_params = set()  # ids of parameters a launch of the CHAIN itself accumulated into (complete with the last segment only)
        self.mode = "relaxed"      # capture error mode: segments are closed / opened from inside autograd's backward

    # ---- recording ----
    def _begin_segment(self):
        self.cur = torch.cuda.CUDAGraph()
        self.cur.capture_begin(pool=self.pool, capture_error_mode=self.mode)

    def _end_segment(self):
        g, self.cur = self.cur, None
        g.capture_end()
        self.chain.append(g)

    cut_pending = False

    def fork(self, k, fn, dead_end=False):
        self.jobs[-1].append((k, fn))
        self.nforks += 1
        if dead_end or len(self.jobs[-1]) >= self.K:
            # the segment is closed in front of the chain's NEXT library launch (ops._stream calls do_cut): a cut behind the
            # last launch of backward would leave an empty graph, and torch ops in between (gradient accumulation) stay with
            # the segment that produced their operands
            self.cut_pending = True

    def do_cut(self):
        self.cut_pending = False
        self._end_segment()
        self.jobs.append([])
        self._begin_segment()

    def finish(self):
        """Called by ops.OVERLAP.end() right behind loss.backward(), still inside ops.direct_param_grads(): close the last
        chain segment, then record the side graphs."""
        self.cut_pending = False
        self._end_segment()
        from . import ops

        streams = ops.OVERLAP.streams
        listener = _DoneRecorder()
        prev_listener, ops.GRAD_LISTENER = ops.GRAD_LISTENER, listener
        try:
            for seg in self.jobs:
                graphs = {}
                listener.cur = set()
                for k in sorted({k for k, _ in seg}):
                    with torch.cuda.stream(streams[k]):
                        g = torch.cuda.CUDAGraph()
                        g.capture_begin(pool=self.side_pool, capture_error_mode=self.mode)
                        try:
                            for kk, fn in seg:
                                if kk == k:
                                    fn()
                        finally:
                            g.capture_end()
                    graphs[k] = g
                self.side.append(graphs)
                self.events.append(torch.cuda.Event())
                self.done_events.append({k: torch.cuda.Event() for k in graphs})
                self.seg_params.append(listener.cur)
        finally:
            ops.GRAD_LISTENER = prev_listener
        self.jobs = None   # the closures (and what they reference) are not needed again
        last = {}          # the join: behind the LAST graph of every side stream that runs one
        for i, graphs in enumerate(self.side):
            for k in graphs:
                last[k] = self.done_events[i][k]
        self.join_events = list(last.values())

    def abort(self):
        if self.cur is not None:
            try:
                self.cur.capture_end()
            except Exception:
                pass
            self.cur = None

    # ---- replay ----
    def replay(self, between=None):
        """``between(i)``: called behind the launch of segment ``i``'s side graphs (the trainer launches gradient buckets that
        segment completes from it)."""
        from . import ops

        tr = self.tr
        cs = tr._chain_stream
        streams = ops.OVERLAP.streams
        cur = torch.cuda.current_stream()
        tr._entry_event.record(cur)
        cs.wait_event(tr._entry_event)
        with torch.cuda.stream(cs):
            for i, g in enumerate(self.chain):
                g.replay()
                graphs = self.side[i]
                if graphs or between is not None:
                    self.events[i].record(cs)
                for k, sg in graphs.items():
                    st = streams[k]
                    st.wait_event(self.events[i])
                    with torch.cuda.stream(st):
                        sg.replay()
                        self.done_events[i][k].record(st)
                if between is not None:
                    between(i)
            for ev in self.join_events:
                cs.wait_event(ev)

    def replay_tail(self):
        cs = self.tr._chain_stream
        with torch.cuda.stream(cs):
            if self.tail is not None:
                self.tail.replay()
            self.tr._exit_event.record(cs)
        torch.cuda.current_stream().wait_event(self.tr._exit_event)


class _TouchRecorder:
    """ops.GRAD_LISTENER stand-in inside graphed_training_step's captured backward: the parameters a fused function accumulated
    into its staging view itself (direct mode), so that an untouched parameter reports None, as the eager module does."""

    def __init__(self):
        self.ids = set()

    def note_use(self, params):
        pass

    def note_done(self, params):
        self.ids.update(id(p) for p in params)


class _DoneRecorder:
    """ops.GRAD_LISTENER stand-in while the side graphs are recorded: which parameters each segment's side work completes."""

    def __init__(self):
        self.cur = set()

    def note_use(self, params):
        pass

    def note_done(self, params):
        self.cur.update(id(p) for p in params)


class Trainer:
    """``loss = module(*batch)[-1]``; backward; bucketed all-reduce; AdamW.

    ``optimizer_factory(flat_param, flat_grad) -> object with .step(grad_scale)``
    defaults to the HIP AdamW kernel (GPU only; there is no CPU optimizer in the
    product -- tests that exercise the collective logic on CPU/gloo inject one).
    """

    def __init__(self, module: nn.Module, lr=1e-3, betas=(0.9, 0.95), weight_decay=1e-2, eps=1e-8,
                 bucket_bytes: int = 32 << 20, optimizer_factory=None, group=None, use_graph: bool = False,
                 overlap_wgrad: bool = True, early_leaf_backward: bool | None = None, grad_comm_dtype=None,
                 executor: str | None = None, forks_per_segment: int | None = None):
        import os

        self.module = module
        self.use_graph = use_graph
        if os.environ.get("NLAM_OVERLAP_WGRAD") in ("0", "1"):   # A/B runs: the whole step on one stream (0)
            overlap_wgrad = os.environ["NLAM_OVERLAP_WGRAD"] == "1"
        self.overlap_wgrad = overlap_wgrad
        # how a captured step is replayed: "segments" = a chain of linear graphs on one stream + weight-gradient
        # graphs on side streams (_SegmentedStep); "forks" = ONE graph whose weight-gradient branches the executor places
        # "auto" (default): measured, round 5 (profiles/round5/ab_segmented_executor_wide.log, same box each): the segmented
        # executor wins where the kernels are long against a graph-launch boundary -- cfg3 (d = 256) 46.8 -> 43.7 ms, cfg5
        # (d = 512) 139.7 -> 129.8 ms, cfg4 (Hi-LAM d = 128) 11.18 -> 10.72 ms -- and loses where they are not: cfg2 (d = 64)
        # 1.75 -> 1.94 ms, cfg4p (chunked Hi-LAM-Parallel) 6.43 -> 6.51 ms.  So: segments for modules with a fused width above 64
        # and no chunked (SplitMLPs) stage, the one-graph executor otherwise.
        self.executor = executor or os.environ.get("NLAM_EXEC", "auto")
        if self.executor not in ("segments", "forks", "auto"):
            raise ValueError(f"unknown executor {self.executor!r}: 'segments', 'forks' or 'auto'")
        if self.executor == "auto":
            self.executor = self._pick_executor(module)
        self.forks_per_segment = int(forks_per_segment if forks_per_segment is not None else os.environ.get("NLAM_SEG_FORKS", "12"))
        if early_leaf_backward is None:
            early_leaf_backward = os.environ.get("NLAM_EARLY_LEAF", "0") == "1"
        self.early_leaf_backward = early_leaf_backward
        self._graph = None
        self._static_in = None
        self._static_sig = None
        self._static_loss = None
        self.fp = FlatParams(module)
        self.buckets = GradBuckets(self.fp, bucket_bytes, group, comm_dtype=grad_comm_dtype)
        self.buckets.join_streams = self._join_producers
        self.world = self.buckets.world
        if optimizer_factory is None:
            from .ops import AdamWFlat

            self.opt = AdamWFlat(self.fp.flat, self.fp.grad, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        else:
            self.opt = optimizer_factory(self.fp.flat, self.fp.grad)

    @staticmethod
    def _pick_executor(module) -> str:
        wide = any(p.dim() == 2 and p.shape[0] > 64 for p in module.parameters())
        chunked = any(type(m).__name__ == "SplitMLPs" for m in module.modules())
        return "segments" if (wide and not chunked) else "forks"

    # ---- HIP-graph step: zero-grad + forward + loss + backward are ~330 launches of 3-150 us at cfg2,
    # which eager Python cannot issue as fast as the GPU retires them; captured once, replayed per step ----
    def _ops(self):
        try:
            from . import ops

            return ops
        except Exception:   # CPU-only test environments without the HIP library
            return None

    def _join_producers(self):
        """Make the current stream wait for every stream that wrote gradients (weight-gradient side streams + the
        backward stream) before a collective is launched from it."""
        ops = self._ops()
        if ops is None or not torch.cuda.is_available():
            return
        cur = torch.cuda.current_stream()
        for st in [*ops.OVERLAP.streams, self._main_stream]:
            if st is not None and st != cur:
                cur.wait_stream(st)

    _main_stream = None
    _packer = None           # ops.WeightPacker: weight images of the narrow fused kernels, rewritten once per step
    _unit = None
    _module_kwargs = {}      # extra keyword arguments of the module call inside the captured step
    _pre_standardize = False

    def _fwd_bwd_on(self, batch):
        """forward + backward.  For their duration the fused-MLP backward owns the parameter gradients
        (``ops.direct_param_grads``: partial sums reduced straight into the flat views, bucket completion reported to
        ``self.buckets``) and the weight-gradient kernels are forked onto side streams (``ops._WgradOverlap``)."""
        ops = self._ops()
        on_gpu = ops is not None and self.fp.flat.is_cuda
        if not on_gpu:
            out = self.module(*batch)
            loss = out[-1] if isinstance(out, tuple) else out
            loss.backward()
            return loss.detach()
        self._main_stream = torch.cuda.current_stream()
        if self._packer is None:
            self._packer = ops.WeightPacker()
        with ops.direct_param_grads(self.buckets, early_leaf=self.early_leaf_backward, packer=self._packer):
            out = self.module(*batch, **self._module_kwargs)
            loss = out[-1] if isinstance(out, tuple) else out
            ov = ops.OVERLAP if self.overlap_wgrad else None
            seg = self._recording   # a _SegmentedStep being recorded: the forks are handed to it instead of being launched
            if seg is not None and ov is None:
                raise RuntimeError("the segmented executor records the weight-gradient work through ops.OVERLAP (overlap_wgrad=True)")
            if ov is not None:
                ov.begin(deferred=seg)
            # a recording closes / opens stream captures from inside backward: autograd must run it on THIS thread
            threads = torch.autograd.set_multithreading_enabled(False) if seg is not None else contextlib.nullcontext()
            try:
                with threads:
                    if loss.dim() == 0 and loss.dtype == torch.float32:
                        if self._unit is None or self._unit.device != loss.device:
                            self._unit = torch.ones((), device=loss.device, dtype=torch.float32)
                        loss.backward(gradient=self._unit)   # a resident seed: autograd's implicit ones_like is a fill launch per step
                    else:
                        loss.backward()
            except BaseException:
                if seg is not None:
                    ov.deferred = None   # no side graphs for a failed recording
                    ov.active = False
                raise
            finally:
                if ov is not None:
                    ov.end()
        return loss.detach()   # nothing that references the autograd graph survives this frame

    _recording = None

    def _fwd_bwd(self):
        return self._fwd_bwd_on(self._static_in)

    def _capture(self, *batch):
        # No cyclic garbage collection between here and the end of the capture: on ROCm the destructor of a torch CUDAGraph
        # synchronises the device, which is an error inside a stream capture and, thrown from a destructor, aborts the
        # process -- an older Trainer that the collector happens to free from the autograd thread mid-capture did that.
        import gc

        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            self._capture_locked(*batch)
        finally:
            if was_enabled:
                gc.enable()

    def _capture_locked(self, *batch):
        # A module that standardises its inputs (ForecasterStep(standardize=True)) does so OUTSIDE the captured step, from
        # the caller's batch straight into the graph's input buffers: the staging copy of the batch (17 MB at cfg2, one
        # multi-tensor launch of 22 us) and on_after_batch_transfer become one launch.  Same kernel, same bits.
        self._pre_standardize = (len(batch) == 3 and bool(getattr(self.module, "standardize_inputs", False))
                                 and hasattr(self.module, "standardize") and all(b.is_cuda and b.dtype == torch.float32 for b in batch))
        if self._pre_standardize:
            self._static_in = [torch.empty_like(b, memory_format=torch.contiguous_format) for b in batch]
            self.module.standardize(*batch, out=self._static_in)
            self._module_kwargs = {"standardize": False}
        else:
            self._static_in = [b.clone() for b in batch]
        self._static_sig = [(tuple(b.shape), b.dtype) for b in batch]
        self.buckets.enabled = False
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):   # warm-up outside capture: lazy CSR builds, LDS-size attributes, allocator pool
            for _ in range(2):
                self.fp.zero_grad()
                self._fwd_bwd()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        # the warm-up's autograd graph (and its AccumulateGrad nodes, bound to the side stream) is gone here, so the
        # capture creates its own on the capture stream and accumulates in place into the flat gradient views
        t_before = getattr(self.opt, "t", None)
        if self.executor == "segments" and self.overlap_wgrad:
            self._capture_segments(t_before)
            return
        self._graph = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread may query events while this thread captures
        # world == 1: nothing sits between backward and the optimizer, so AdamW (step count resident on the device) is captured
        # too -- one launch latency less per step than enqueueing it behind the replay.  With data parallelism the gradient
        # all-reduce separates the two and stays outside the capture.
        self._opt_in_graph = self.world == 1 and bool(getattr(self.opt, "capturable", False)) and not self._opt_eager
        self._tail_graph = None
        try:
            with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                self.fp.grad.zero_()
                self._static_loss = self._fwd_bwd()
                if self._opt_in_graph:
                    self.opt.step(1.0)
            self._capture_tail()
        finally:
            if t_before is not None:
                self.opt.t = t_before   # the capture recorded the launches without running them (also when it failed half way)
        self._opt_sig = self._opt_signature()

    _tail_graph = None

    def _capture_tail(self):
        """world > 1, one-graph executor: the gradient all-reduce separates backward from the optimizer, so the optimizer is a
        second graph, replayed behind the collective (``1 / world`` folded into the captured AdamW launch; step count and bias
        corrections live on the device) -- no per-step launch arguments, one replay instead of two eager launches."""
        self._tail_graph = None
        if self.world > 1 and bool(getattr(self.opt, "capturable", False)) and not self._opt_eager:
            t_before = getattr(self.opt, "t", None)
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, capture_error_mode="thread_local"):
                    self.opt.step(1.0 / self.world)
            finally:
                if t_before is not None:
                    self.opt.t = t_before
            self._tail_graph = g

    _chain_stream = None
    _entry_event = None
    _exit_event = None
    _opt_eager = False     # the optimizer's hyper-parameters keep changing (a schedule): it runs behind the replay, uncaptured
    _bucket_plan = None

    def _capture_segments(self, t_before):
        """Record the step as a _SegmentedStep (see its docstring): chain segments on one stream, one
        weight-gradient graph per (segment, side stream), the optimizer as a last graph behind the join."""
        import os

        if self._chain_stream is None:
            # normal priority: a HIGH-priority chain stream does get a hardware queue of its own, but the queue priority starves
            # and context-switches the side queues instead of sharing the chip with them: 4.0-4.9 ms per cfg2 step against 1.94
            # a stream bound to a hardware queue that no weight-gradient side stream uses (ops.stream_layout: observed, once per process)
            from . import ops

            prio = int(os.environ.get("NLAM_CHAIN_PRIO", "0"))
            self._chain_stream = ops.stream_layout()["chain"] if (prio == 0 and ops.QUEUE_SIDES != "0") else torch.cuda.Stream(priority=prio)
            self._entry_event, self._exit_event = torch.cuda.Event(), torch.cuda.Event()
        cs = self._chain_stream
        seg = _SegmentedStep(self, self.forks_per_segment)
        # the optimizer is captured whenever it can be -- at world > 1 too: the gradient all-reduce runs between the join
        # and its graph, with 1 / world folded into the captured AdamW launch
        self._opt_in_graph = bool(getattr(self.opt, "capturable", False)) and not self._opt_eager
        cs.wait_stream(torch.cuda.current_stream())
        try:
            with torch.cuda.stream(cs):
                self._recording = seg
                seg._begin_segment()
                # (the side closures run under their own listener inside seg.finish(): what reaches self.buckets here are the
                # reductions a chain launch did itself)
                self.buckets.chain_recorder = seg.chain_params
                try:
                    self.fp.grad.zero_()
                    self._static_loss = self._fwd_bwd()   # ops.OVERLAP.end() -> seg.finish(): every graph but the optimizer's
                except BaseException:
                    seg.abort()
                    raise
                finally:
                    self._recording = None
                    self.buckets.chain_recorder = None
                if self._opt_in_graph:
                    seg.tail = torch.cuda.CUDAGraph()
                    seg.tail.capture_begin(pool=seg.pool, capture_error_mode=seg.mode)
                    try:
                        self.opt.step(1.0 / self.world)
                    finally:
                        seg.tail.capture_end()
        finally:
            if t_before is not None:
                self.opt.t = t_before
        torch.cuda.current_stream().wait_stream(cs)
        self._graph = seg
        self._bucket_plan = self._plan_buckets(seg) if self.world > 1 else None
        self._opt_sig = self._opt_signature()

    def _plan_buckets(self, seg):
        """world > 1: per chain segment the gradient buckets whose collective is launched behind it.  A bucket is complete when
        the side work of the last segment that reported one of its parameters has run (a rollout back-propagates a parameter
        once per AR step: the last report counts); a parameter no side graph reported -- accumulated by autograd, or by a
        kernel of the chain itself -- is complete with the last segment.  Buckets go out in index order on every rank, so one
        that is ready late holds back the ones behind it."""
        nseg = len(seg.chain)
        last_seg = {}
        for i, ids in enumerate(seg.seg_params):
            for pid in ids:
                last_seg[pid] = i
        for pid in getattr(seg, "chain_params", ()):   # also written by a chain launch (a later segment may add to it): ready with the last segment
            last_seg[pid] = nseg - 1
        out, cur = [[] for _ in range(nseg)], 0
        for b, (_s0, _e0, members) in enumerate(self.buckets.bounds):
            ready = max((last_seg.get(id(self.fp.params[i]), nseg - 1) for i in members), default=nseg - 1)
            cur = max(cur, ready)
            out[cur].append(b)
        return out

    def _opt_signature(self):
        """The optimizer's hyper-parameters are launch arguments of the captured AdamW kernel: a change (a learning-rate
        schedule) must re-record it, or the replays would keep applying the old values."""
        o = self.opt
        captured = self._opt_in_graph or self._tail_graph is not None
        return tuple(getattr(o, k, None) for k in ("lr", "betas", "eps", "wd")) if captured else None

    _opt_sig = None
    _opt_changes = 0

    def _optimizer_changed(self):
        """lr / betas / eps / weight decay differ from what the captured AdamW launch carries.  The first change re-records
        (segmented executor: the optimizer's own two-launch graph; one-graph executor: the whole step).  A second change is a
        schedule: from then on the optimizer runs uncaptured behind the replay -- two launches per step instead of a
        re-recording per step (advisor finding, round 4)."""
        self._opt_changes += 1
        seg = self._graph if isinstance(self._graph, _SegmentedStep) else None
        if self._opt_changes >= 2:
            self._opt_eager = True
            if seg is not None:
                seg.tail = None
            elif self._opt_in_graph:
                self._graph = None   # recorded again, once, without the optimizer
            self._opt_in_graph = False
            self._tail_graph = None
            self._opt_sig = None
            return
        # re-recording is a stream capture: no cyclic garbage collection inside it (see _capture)
        import gc

        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()
        try:
            if seg is None:
                if self._tail_graph is not None:   # the optimizer's own graph: only that is recorded again
                    self._capture_tail()
                    self._opt_sig = self._opt_signature()
                else:
                    self._graph = None
                return
            t_before = getattr(self.opt, "t", None)
            try:
                with torch.cuda.stream(self._chain_stream):
                    seg.tail = torch.cuda.CUDAGraph()
                    seg.tail.capture_begin(pool=seg.pool, capture_error_mode=seg.mode)
                    try:
                        self.opt.step(1.0 / self.world)
                    finally:
                        seg.tail.capture_end()
            finally:
                if t_before is not None:
                    self.opt.t = t_before   # the capture recorded the launches without running them (also when it failed)
            self._opt_sig = self._opt_signature()
        finally:
            if was_enabled:
                gc.enable()

    def _graph_step(self, *batch):
        if self._graph is not None and (self._opt_in_graph or self._tail_graph is not None) and self._opt_signature() != self._opt_sig:
            self._optimizer_changed()
        if self._graph is None:
            try:
                self._capture(*batch)
            except Exception as exc:  # capture is an optimisation: never let it take the job down
                import warnings

                warnings.warn(f"HIP-graph capture of the training step failed ({exc!r}); continuing with eager launches")
                self._graph = None
                self.use_graph = False
                self._opt_in_graph = False
                self._module_kwargs, self._pre_standardize = {}, False
                torch.cuda.synchronize()
                return self.step(*batch)
        sig = [(tuple(b.shape), b.dtype) for b in batch]
        if sig != self._static_sig:
            # a HIP graph is one shape: a batch of another shape (a last partial batch, say) takes the eager step --
            # copying it into the captured buffers would broadcast or fail, and the replay would answer for the wrong batch
            self.use_graph = False
            kw, self._module_kwargs = self._module_kwargs, {}
            try:
                return self.step(*batch)
            finally:
                self.use_graph = True
                self._module_kwargs = kw
        if self._pre_standardize:
            self.module.standardize(*batch, out=self._static_in)
        else:
            torch._foreach_copy_(self._static_in, list(batch))   # one multi-tensor launch instead of one copy per input
        self._replay_step()
        return self._static_loss.clone()   # the static tensor is overwritten by the next replay

    _opt_in_graph = False
    _comm_stream = None
    bucket_launch_segments = None   # world > 1, segmented executor: (bucket, chain segment behind which it was launched), last step

    def _replay_step(self):
        """One replay of the recorded step + gradient exchange + optimizer."""
        g = self._graph
        if not isinstance(g, _SegmentedStep):
            g.replay()
            self._after_replay()
            return
        if self.world > 1:
            # bucket collectives are launched from a communication stream that waits, on the device, for the chain segment and
            # the weight-gradient graphs completing the bucket; the chain goes on meanwhile.  RCCL calls stay outside every
            # capture: a communicator fault can never poison a graph.
            if self._comm_stream is None:
                from . import ops

                placed = ops.stream_layout().get("comm") if ops.QUEUE_SIDES != "0" else None
                self._comm_stream = placed if placed is not None else torch.cuda.Stream()
            comm, bk = self._comm_stream, self.buckets
            bk.handles, bk.launched = [], []
            self.bucket_launch_segments = []

            def between(i):
                comm.wait_event(g.events[i])
                for ev in g.done_events[i].values():
                    comm.wait_event(ev)
                for b in self._bucket_plan[i]:
                    s0, e0, _ = bk.bounds[b]
                    with torch.cuda.stream(comm):
                        bk._reduce_slice(s0, e0)
                    bk.launched.append(b)
                    self.bucket_launch_segments.append((b, i))

            g.replay(between=between)
            with torch.cuda.stream(self._chain_stream):
                bk._wait_all()    # the chain stream waits for the collectives (and writes back a reduced-precision exchange)
        else:
            g.replay()
        g.replay_tail()           # the optimizer's graph (if recorded) on the chain stream; the caller's stream waits for it
        if self._opt_in_graph:
            self.opt.note_replayed()
        else:
            self.opt.step(1.0 / self.world)

    def _after_replay(self):
        """One-graph executor: gradient exchange + optimizer behind the replay (the optimizer is part of the graph when
        world == 1)."""
        if self._opt_in_graph:
            self.opt.note_replayed()
            return
        if self.world > 1:   # one flat buffer: a single collective (0.86 MB at cfg2, 20.6 MB at cfg3; half of that with bf16 exchange)
            self.buckets.all_reduce_whole()
        if self._tail_graph is not None:
            self._tail_graph.replay()
            self.opt.note_replayed()
        else:
            self.opt.step(1.0 / self.world)

    def step(self, *batch):
        if self.use_graph:
            return self._graph_step(*batch)
        self.buckets.enabled = True
        self.fp.zero_grad()
        self.buckets.begin_step()
        loss = self._fwd_bwd_on(batch)
        self.buckets.finish_step()
        self.opt.step(1.0 / self.world)
        return loss

    def step_from(self, dataset, indices, standardize=True):
        """One optimizer step on ``dataset.batch(indices)`` (``neural_lam_amd.data.DeviceWeatherDataset``): the data path of
        the reference's training loop -- WeatherDataset.__getitem__ + collation + on_after_batch_transfer,
        weather_dataset.py:467-533, models/module.py:326-367 -- as ONE launch in front of the step.  With a captured
        step the samples are written straight into the graph's static input buffers (no staging copy); ``indices`` may
        be a slice of a device-resident permutation, so an epoch runs without host->device traffic.

        ``standardize=True`` folds on_after_batch_transfer into that launch: the module must then NOT standardise again
        (``ForecasterStep(standardize=False)``).  Returns the loss; ``self.batch_times`` holds the target times."""
        if getattr(self.module, "standardize_inputs", False) and standardize:
            raise ValueError("the module standardises its inputs itself: pass standardize=False or build it with standardize=False")
        if self.use_graph and self._graph is not None:
            B = int(indices.numel()) if isinstance(indices, torch.Tensor) else len(indices)
            if self._static_in[0].shape[0] == B and len(self._static_in) == 3:
                if self.batch_times is None or self.batch_times.shape != (B, dataset.ar_steps):
                    self.batch_times = torch.empty((B, dataset.ar_steps), device=self._static_in[0].device, dtype=torch.int64)
                if self._pre_standardize:
                    # the captured step was recorded WITHOUT on_after_batch_transfer (it is hoisted out of the capture into
                    # module.standardize(..., out=static inputs)): cut the raw batch, then standardise it into the graph's
                    # input buffers -- writing the raw samples there would train on unstandardised data
                    raw = dataset.batch(indices, standardize=False)
                    self.module.standardize(*raw[:3], out=self._static_in)
                    self.batch_times = raw[3]
                else:
                    dataset.batch(indices, standardize=standardize, out=(*self._static_in, self.batch_times))
                self._replay_step()
                return self._static_loss.clone()
        init, target, forcing, self.batch_times = dataset.batch(indices, standardize=standardize)
        return self.step(init, target, forcing)

    batch_times = None


# ---------------------------------------------------------------------------
# The drop-in path: a captured forward + backward for a training loop that owns its optimizer (Lightning)
# ---------------------------------------------------------------------------
class _GraphedStep:
    """See ``graphed_training_step``."""

    def __init__(self, module: nn.Module, sample_args, warmup: int = 3, pack_weights: bool = True, overlap_wgrad: bool = True,
                 flat: bool = False):
        from . import ops

        if not all(isinstance(a, torch.Tensor) and a.is_cuda for a in sample_args):
            raise ValueError("graphed_training_step: the sample arguments must be tensors on the GPU")
        self.module = module
        self.params = tuple(p for p in module.parameters() if p.requires_grad)
        if not self.params:
            raise ValueError("module has no trainable parameters")
        self.flat_parameter = None
        if flat:
            if not overlap_wgrad:
                raise ValueError("graphed_training_step(flat=True) needs overlap_wgrad=True (the fused MLPs write the flat gradient)")
            if any(p.dtype != torch.float32 or p.device != self.params[0].device for p in self.params):
                raise ValueError("graphed_training_step(flat=True): every trainable parameter must be fp32 and on one device")
        self.autocast = (torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda"))
        self.static_in = [a.detach().clone().requires_grad_(a.requires_grad) for a in sample_args]
        self.sig = [(tuple(a.shape), a.dtype, a.requires_grad) for a in sample_args]
        self.packer = ops.WeightPacker() if pack_weights else None
        surface = tuple(a for a in self.static_in if a.requires_grad) + self.params
        self.n_in_grads = sum(1 for a in self.static_in if a.requires_grad)
        # overlap_wgrad: inside the captured backward the fused MLPs own the parameter gradients the way they do under
        # trainer.Trainer -- partial sums reduced straight into views of ONE static staging buffer, the weight-gradient kernels
        # forked onto side streams (parallel branches of the backward graph) -- and the staging views are what the replay hands to
        # autograd as the parameters' gradients.  Same bits as the eager module (0 + x = x; accumulation order = backward order).
        self.gflat, self.gviews = None, None
        if overlap_wgrad:
            offs, off = [], 0
            for p in self.params:
                offs.append(off)
                off += (p.numel() + 3) // 4 * 4
            self.gflat = torch.zeros(off, device=self.params[0].device, dtype=torch.float32)
            self.goffs = offs
            self.gviews = [self.gflat[o : o + p.numel()].view(p.shape) for o, p in zip(offs, self.params)]
            if flat:
                # ONE leaf for autograd and the optimizer: the module's parameters become views of one flat buffer laid out like
                # the gradient staging buffer (names, shapes and state_dict unchanged: FlatParams does the same for Trainer), and
                # the leaf the replay function takes -- and the caller's optimizer steps -- is a Parameter over that storage
                pflat = torch.zeros(off, device=self.params[0].device, dtype=torch.float32)
                with torch.no_grad():
                    for o, p in zip(offs, self.params):
                        pflat[o : o + p.numel()].copy_(p.data.reshape(-1))
                        p.data = pflat[o : o + p.numel()].view(p.shape)
                self.flat_parameter = nn.Parameter(pflat)

        import gc

        def forward():
            prev, ops.PACKER = ops.PACKER, self.packer
            try:
                if self.packer is not None:
                    self.packer.begin_step()   # the weight images, rewritten from the current weights: first kernels of the graph
                out = module(*self.static_in)
            finally:
                ops.PACKER = prev
            return out if isinstance(out, tuple) else (out,)

        def backward(outs, gouts, direct=False):
            live = [(o, g) for o, g in zip(outs, gouts) if g is not None]
            prev, ops.PACKER = ops.PACKER, self.packer
            shape = ops.wgrad_shape("1")   # the eager module's weight-gradient launch shape: same row slices, same sums, bit for bit
            shape.__enter__()
            try:
                if not direct:
                    return torch.autograd.grad(tuple(o for o, _ in live), surface, grad_outputs=tuple(g for _, g in live),
                                               only_inputs=True, allow_unused=True)
                fp32 = [p.dtype == torch.float32 for p in self.params]
                held = [p.grad for p in self.params]
                state = (ops.DIRECT_PARAM_GRADS, ops.GRAD_LISTENER, ops.EARLY_LEAF_BACKWARD)
                self.gflat.zero_()
                for p, v, ok in zip(self.params, self.gviews, fp32):
                    if ok:
                        p.grad = v
                rec = _TouchRecorder()
                ops.DIRECT_PARAM_GRADS, ops.GRAD_LISTENER, ops.EARLY_LEAF_BACKWARD = True, rec, False
                ops.OVERLAP.begin()
                try:
                    got = torch.autograd.grad(tuple(o for o, _ in live), surface, grad_outputs=tuple(g for _, g in live),
                                              only_inputs=True, allow_unused=True)
                finally:
                    ops.OVERLAP.end()   # the side streams join the capture here
                    ops.DIRECT_PARAM_GRADS, ops.GRAD_LISTENER, ops.EARLY_LEAF_BACKWARD = state
                    for p, g in zip(self.params, held):
                        p.grad = g
                got = list(got)
                touched = rec.ids
                for i, (p, v, ok) in enumerate(zip(self.params, self.gviews, fp32)):
                    k = self.n_in_grads + i
                    # what the fused MLPs accumulated themselves (+ whatever came back through autograd for this parameter); a
                    # parameter neither of them produced keeps None, as under the eager module (torch AdamW / DDP skip it)
                    if self.flat_parameter is not None:
                        # flat mode: whatever autograd delivered for a parameter is added onto its slice of the staging buffer
                        # (captured); the buffer as a whole is the gradient of the flat leaf
                        if got[k] is not None:
                            v.add_(got[k])
                        got[k] = None
                    elif ok and id(p) in touched:
                        got[k] = v if got[k] is None else v + got[k]
                return tuple(got)
            finally:
                shape.__exit__(None, None, None)
                ops.PACKER = prev

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):   # eager passes first: lazy graph layouts, LDS attributes, packer tables, allocator state
            for _ in range(max(1, warmup)):
                outs = forward()
                backward(outs, [torch.zeros_like(o) if o.requires_grad else None for o in outs])
            del outs
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gc.collect()
        was_enabled = gc.isenabled()
        gc.disable()   # a CUDAGraph destructor synchronises the device: never inside a capture (see Trainer._capture)
        try:
            self.pool = torch.cuda.graph_pool_handle()
            self.fwd_graph, self.bwd_graph = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.fwd_graph, pool=self.pool, capture_error_mode="thread_local"):
                self.static_out = forward()
            self.static_gout = [torch.zeros_like(o) if o.requires_grad else None for o in self.static_out]
            with torch.cuda.graph(self.bwd_graph, pool=self.pool, capture_error_mode="thread_local"):
                self.static_grads = backward(self.static_out, self.static_gout, direct=self.gflat is not None)
        finally:
            if was_enabled:
                gc.enable()
        outer = self

        class _Replay(torch.autograd.Function):
            @staticmethod
            def forward(ctx, *flat):
                for dst, src in zip(outer.static_in, flat[: len(outer.static_in)]):
                    if dst.data_ptr() != src.data_ptr():
                        dst.copy_(src)
                outer.fwd_graph.replay()
                return tuple(o.detach() for o in outer.static_out)

            @staticmethod
            @torch.autograd.function.once_differentiable
            def backward(ctx, *gouts):
                for dst, g in zip(outer.static_gout, gouts):
                    if dst is not None:
                        if g is None:
                            dst.zero_()
                        elif dst.data_ptr() != g.data_ptr():
                            dst.copy_(g)
                outer.bwd_graph.replay()
                if outer.flat_parameter is not None:   # one gradient for one leaf (a copy: AccumulateGrad keeps what it is handed)
                    it = iter(g.detach().clone() if g is not None else None for g in outer.static_grads[: outer.n_in_grads])
                    return (*[next(it) if a.requires_grad else None for a in outer.static_in], outer.gflat.clone())
                if outer.gflat is not None:
                    # the staging buffer is copied once (one launch) and autograd gets views of the COPY: AccumulateGrad keeps
                    # (steals) what it is handed as ``.grad``, and a ``.grad`` that aliased the staging buffer would be added
                    # to itself by the next backward under ``zero_grad(set_to_none=False)``
                    fresh = outer.gflat.clone()
                    grads = []
                    for k, g in enumerate(outer.static_grads):
                        i = k - outer.n_in_grads
                        if i >= 0 and g is outer.gviews[i]:
                            grads.append(fresh[outer.goffs[i] : outer.goffs[i] + g.numel()].view(g.shape))
                        else:
                            grads.append(g.detach().clone() if g is not None else None)
                else:   # torch.cuda.make_graphed_callables' contract: the static gradient tensors themselves
                    grads = [g.detach() if g is not None else None for g in outer.static_grads]
                it = iter(grads)
                res = [next(it) if a.requires_grad else None for a in outer.static_in]
                return (*res, *it)

        self._fn = _Replay

    def __call__(self, *args):
        sig = [(tuple(a.shape), a.dtype, a.requires_grad) for a in args]
        ac = (torch.is_autocast_enabled("cuda"), torch.get_autocast_dtype("cuda"))
        if sig != self.sig or (ac[0] != self.autocast[0]) or (ac[0] and ac[1] != self.autocast[1]):
            if self.flat_parameter is not None and torch.is_grad_enabled():
                # the eager module would deliver its gradients to the individual parameters, which the caller's optimizer
                # (built over ``flat_parameter``) never sees: refuse rather than train on nothing
                raise ValueError("graphed_training_step(flat=True): batch signature %r differs from the captured %r; under grad mode "
                                 "only the captured shape / autocast state is accepted" % (sig, self.sig))
            return self.module(*args)   # another batch shape / autocast state: the module itself (eager launches)
        if self.flat_parameter is not None:
            out = self._fn.apply(*args, self.flat_parameter)
        else:
            out = self._fn.apply(*args, *self.params)
        return out if len(out) > 1 else out[0]


class FlatStepModule(nn.Module):
    """A captured step (``graphed_training_step(..., flat=True)``) as an ``nn.Module`` whose ONLY parameter is the flat leaf:
    what ``torch.nn.parallel.DistributedDataParallel`` wraps (one parameter, one bucket, one hook) and what
    ``torch.optim.AdamW(m.parameters())`` steps.  ``forward(*batch)`` returns what the captured module returns, or element
    ``pick`` of it (DDP wants tensors that need the backward: ``pick=1`` = the loss of ``models.ForecasterStep``)."""

    def __init__(self, step: "_GraphedStep", pick=None):
        super().__init__()
        if step.flat_parameter is None:
            raise ValueError("FlatStepModule needs graphed_training_step(..., flat=True)")
        self.flat = step.flat_parameter
        self._step = (step,)   # not a registered submodule: the captured module's own parameters are views of ``flat``
        self.pick = pick

    def forward(self, *batch):
        out = self._step[0](*batch)
        return out if self.pick is None else out[self.pick]


def graphed_training_step(module: nn.Module, *sample_args, warmup: int = 3, pack_weights: bool = True, overlap_wgrad: bool = True,
                          flat: bool = False):
    """``module`` (e.g. ``models.ForecasterStep``: batch -> (prediction, loss)) as a callable whose forward AND backward each
    replay one HIP graph -- for a training loop that keeps its own optimizer and gradient handling, i.e. the reference's:
    ``ForecasterModule.training_step`` called by ``pl.Trainer`` (models/module.py:394-417, train_model.py:564-578) returns the
    loss, Lightning calls ``loss.backward()`` and ``torch.optim.AdamW.step()``, DDP's hooks see every ``.grad`` arrive.

        step = graphed_training_step(forecaster_step, init, target, forcing)     # once, on a sample batch (it is not consumed)
        ...
        def training_step(self, batch):                                          # LightningModule
            prediction, loss = step(*batch)
            return loss

    The same contract as ``torch.cuda.make_graphed_callables`` (static input / output / gradient buffers; a batch of another
    shape or autocast state falls through to the module's eager launches; parameter gradients come back through autograd, so
    ``AccumulateGrad`` hooks -- DDP -- fire as usual), plus what this library adds: the fused MLPs' weight images are rewritten
    by the first kernels of the forward graph (``pack_weights``), the weight-gradient kernels run on side streams inside the
    backward graph (``overlap_wgrad``), and the capture is thread-local, so a live RCCL watchdog thread does not disturb it.  ~330 launches of 3-150 us per cfg2 step become two graph launches; bench.py reports the step time of
    this path beside the eager one (``lightning_shaped``).  Results are bit-identical to the eager module (test_boundary.py /
    test_hip_parity.py::test_graphed_training_step_equals_eager).

    ``flat=True``: the host side of the loop above is ~130 ``AccumulateGrad`` nodes, ~130 gradient
    views built in Python per backward and a multi-tensor optimizer over ~130 tensors -- more than the GPU work of a cfg2 step.
    With ``flat=True`` the module's parameters are re-homed as views of ONE flat fp32 buffer (names, shapes, ``state_dict``
    unchanged) and autograd sees a single leaf over that buffer, ``step.flat_parameter``: one ``AccumulateGrad``, one
    gradient (the staging buffer the fused MLPs reduce into, copied once), and the caller builds its optimizer over that leaf,

        step = graphed_training_step(forecaster_step, *batch, flat=True)
        opt = torch.optim.AdamW([step.flat_parameter], lr=1e-3, betas=(0.9, 0.95))     # models/module.py:293-304

    AdamW is element-wise and the reference puts every parameter in one group (no per-tensor weight decay exceptions), so the
    update is the one ``AdamW(module.parameters())`` makes, bit for bit (test_graphed_flat_step_equals_eager); a parameter
    the step never touches gets a zero gradient instead of ``None`` (it decays; the shipped models have none).  The individual
    parameters get no ``.grad``: under DDP wrap ``FlatStepModule(step, pick=1)`` (one parameter, one bucket).  Under grad mode
    only the captured batch shape is accepted (an eager fall-through would train parameters the optimizer does not hold)."""
    return _GraphedStep(module, sample_args, warmup=warmup, pack_weights=pack_weights, overlap_wgrad=overlap_wgrad, flat=flat)

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

    def note_done(self, params):
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


class Trainer:
    """``loss = module(*batch)[-1]``; backward; bucketed all-reduce; AdamW.

    ``optimizer_factory(flat_param, flat_grad) -> object with .step(grad_scale)``
    defaults to the HIP AdamW kernel (GPU only; there is no CPU optimizer in the
    product -- tests that exercise the collective logic on CPU/gloo inject one).
    """

    def __init__(self, module: nn.Module, lr=1e-3, betas=(0.9, 0.95), weight_decay=1e-2, eps=1e-8,
                 bucket_bytes: int = 32 << 20, optimizer_factory=None, group=None, use_graph: bool = False,
                 overlap_wgrad: bool = True, early_leaf_backward: bool | None = None, grad_comm_dtype=None):
        self.module = module
        self.use_graph = use_graph
        self.overlap_wgrad = overlap_wgrad
        if early_leaf_backward is None:
            import os

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
            if ov is not None:
                ov.begin()
            try:
                if loss.dim() == 0 and loss.dtype == torch.float32:
                    if self._unit is None or self._unit.device != loss.device:
                        self._unit = torch.ones((), device=loss.device, dtype=torch.float32)
                    loss.backward(gradient=self._unit)   # a resident seed: autograd's implicit ones_like is a fill launch per step
                else:
                    loss.backward()
            finally:
                if ov is not None:
                    ov.end()
        return loss.detach()   # nothing that references the autograd graph survives this frame

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
        self._graph = torch.cuda.CUDAGraph()
        # thread_local: the RCCL watchdog thread may query events while this thread captures
        # world == 1: nothing sits between backward and the optimizer, so AdamW (step count resident on the device) is captured
        # too -- one launch latency less per step than enqueueing it behind the replay.  With data parallelism the gradient
        # all-reduce separates the two and stays outside the capture.
        self._opt_in_graph = self.world == 1 and bool(getattr(self.opt, "capturable", False))
        t_before = getattr(self.opt, "t", None)
        try:
            with torch.cuda.graph(self._graph, capture_error_mode="thread_local"):
                self.fp.grad.zero_()
                self._static_loss = self._fwd_bwd()
                if self._opt_in_graph:
                    self.opt.step(1.0)
        finally:
            if t_before is not None:
                self.opt.t = t_before   # the capture recorded the launches without running them (also when it failed half way)
        self._opt_sig = self._opt_signature()

    def _opt_signature(self):
        """The optimizer's hyper-parameters are launch arguments of the captured AdamW kernel: a change (a learning-rate
        schedule) must re-capture, or the replays would keep applying the old values."""
        o = self.opt
        return tuple(getattr(o, k, None) for k in ("lr", "betas", "eps", "wd")) if self._opt_in_graph else None

    _opt_sig = None

    def _graph_step(self, *batch):
        if self._graph is not None and self._opt_in_graph and self._opt_signature() != self._opt_sig:
            self._graph = None   # lr / betas / weight decay changed since the capture: record the step again
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
        self._graph.replay()
        self._after_replay()
        return self._static_loss.clone()   # the static tensor is overwritten by the next replay

    _opt_in_graph = False

    def _after_replay(self):
        """Gradient exchange + optimizer behind a replayed step (the optimizer is part of the graph when world == 1)."""
        if self._opt_in_graph:
            self.opt.note_replayed()
            return
        if self.world > 1:   # one flat buffer: a single collective (0.86 MB at cfg2, 20.6 MB at cfg3; half of that with bf16 exchange)
            self.buckets.all_reduce_whole()
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
                self._graph.replay()
                self._after_replay()
                return self._static_loss.clone()
        init, target, forcing, self.batch_times = dataset.batch(indices, standardize=standardize)
        return self.step(init, target, forcing)

    batch_times = None

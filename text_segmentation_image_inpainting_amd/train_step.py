"""Single-node data-parallel training step for the inpainting and segmentation nets.

One process per GPU.  Images of a minibatch are independent except for BatchNorm batch statistics, which the reference computes
per process-local batch (no SyncBN), so the batch is sharded across ranks with replicated weights and ONE kind of exchange per
iteration: a sum all-reduce of the trainable gradients (RCCL over xGMI through ``torch.distributed``, backend "nccl").

* All trainable parameters, their gradients and momentum buffers live in three flat fp32 buffers laid out in the order gradients
  become ready in backward (reverse registration order), every slice on a 256-byte boundary.  Frozen parameters
  (``mask_conv.weight``, a frozen encoder) never enter them.
* The gradient buffer is cut into buckets (default 16 MB: ImageFill's 26 MB = 2, ImageFillOrigin / V2's 131 / 149 MB = 9 / 10).
  With more than one rank a bucket's all-reduce starts asynchronously from ``post_accumulate_grad`` hooks as soon as its last
  gradient exists, so the exchange overlaps the rest of backward; ``step()`` waits for what is outstanding (``exposed_ms`` measures
  that wait) before the update.  The mean over ranks is folded into the backward seed.
* Collectives must be issued in the same order on every rank, and nothing guarantees that every rank's backward completes its
  buckets in the same order (a rank may recompute a stage the others keep, or build two branches in another order).  The FIRST
  step is therefore a planning step: it exchanges after backward, counts how often each parameter accumulates (a block shared by
  two recomputed segments accumulates twice) and in which order the buckets completed; the counts are agreed over the ranks (max;
  a bucket on whose counts the ranks disagree never overlaps) and rank 0's completion order becomes everybody's LAUNCH order.
  From then on a bucket is complete when each of its parameters has accumulated its planned number of times, and complete buckets
  go out strictly in the agreed order -- a rank whose backward finishes them in another order only waits a little longer.
* The update is the SGD-Nesterov the reference trained with (checkpoints/ReadME.md:4) as one fused HIP kernel over the flat
  buffers; parameters without a gradient keep value and momentum, as under ``torch.optim.SGD``.
* ``capture`` / ``step_graph`` replay forward + backward + packing (+ update on one GPU) from a HIP graph.
"""
import torch
import torch.distributed as dist

from . import ops
from .BaseModels import deferred_batch_counters, to_nhwc


class FlatSGDTrainer:
    """``bucket_mb``: size of the gradient buckets (contiguous ranges of the flat gradient buffer, laid out in the order
    gradients become ready in backward = reverse registration order).  With more than one rank every bucket is
    all-reduced asynchronously as soon as its last gradient has been produced (``post_accumulate_grad`` hooks), so
    the exchange overlaps the rest of the backward pass; ``step()`` waits for the outstanding buckets before the
    update.  ImageFill's 26 MB is two buckets at the default 16 MB, ImageFillOrigin / V2 (131 / 149 MB) nine / ten."""

    def __init__(self, model, lr=0.01, momentum=0.9, weight_decay=1e-4, process_group=None, loss_fn=None, bucket_mb=16.0,
                 overlap=True):
        self.model = model
        named = [p for p in model.parameters() if p.requires_grad]
        self.params = list(reversed(named))          # flat layout = expected gradient-ready order
        self.lr, self.momentum, self.weight_decay = lr, momentum, weight_decay
        self.pg = process_group
        self.distributed = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(process_group) if self.distributed else 1
        # every slice starts on a 64-float (256-byte) boundary: the kernels take their 16-byte vector paths
        # only for aligned operands (an odd-sized tensor, e.g. the [3,35,3,3] head, would misalign the rest)
        pad = lambda n: (n + 63) // 64 * 64
        numel = sum(pad(p.numel()) for p in self.params)
        dev = self.params[0].device
        # flat parameter storage: every trainable parameter becomes a view into one buffer
        self.flat_param = torch.zeros(numel, dtype=torch.float32, device=dev)
        off = 0
        self.slices = []
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat_param[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat_param[off:off + n].view_as(p)
                self.slices.append((off, n))
                off += pad(n)
        self.flat_grad = torch.zeros(numel, dtype=torch.float32, device=dev)   # padding stays 0: no update there
        self.flat_buf = torch.zeros(numel, dtype=torch.float32, device=dev)
        self.grad_views = [self.flat_grad[off:off + n].view_as(p) for p, (off, n) in zip(self.params, self.slices)]
        self.loss_fn = loss_fn or (lambda out_nchw, clean_nhwc: ops.l1_mean(to_nhwc(out_nchw), clean_nhwc))
        # ---- buckets: consecutive parameters up to bucket_mb; bucket b covers flat range [lo, hi)
        limit = max(1, int(bucket_mb * 2**20 / 4))
        self.buckets, cur, lo = [], [], 0
        for i, (o, n) in enumerate(self.slices):
            cur.append(i)
            hi = o + pad(n)
            if hi - lo >= limit or i == len(self.slices) - 1:
                self.buckets.append({"params": cur, "lo": lo, "hi": hi})
                cur, lo = [], hi
        self._bucket_of = {}
        for b, bk in enumerate(self.buckets):
            for i in bk["params"]:
                self._bucket_of[i] = b
        self.overlap = bool(overlap) and self.world > 1
        self._pending, self._works, self._launched, self._fired = None, {}, None, None
        self._skipped = []              # parameters without a gradient in the current step (torch.optim.SGD leaves those alone)
        # the exchange plan (see the module docstring): None until the planning step has run
        self._expected = None           # accumulations per parameter and backward pass, agreed over the ranks
        self._order = None              # the launch order of the overlapped buckets, rank 0's completion order of the planning step
        self._static = set()            # buckets that never overlap (no gradient at all, or the ranks disagree on their counts)
        self._seq, self._last_fire, self._ready, self._next = 0, None, None, 0
        self.deferred_buckets = 0       # buckets off the overlapped path (len(self._static) once planned)
        self.measure_exposed, self._exposed = False, []
        self._hooks = []
        if self.overlap:
            for i, p in enumerate(self.params):
                self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(i)))

    def close(self):
        """Detach from the model: remove the gradient hooks (their closures keep the flat buffers alive and would keep running
        on every backward).  The parameters stay views of ``flat_param``; build the next trainer from the model as usual."""
        for h in self._hooks:
            h.remove()
        self._hooks, self.overlap, self._pending, self._fired = [], False, None, None

    # ---- replication ---------------------------------------------------------------------------------------------
    def broadcast_parameters(self, src=0):
        """Rank ``src``'s weights, optimizer state and module buffers (BatchNorm running statistics and batch
        counters) become every rank's: after a checkpoint load on one rank or a restart all replicas agree."""
        if self.distributed:
            dist.broadcast(self.flat_param, src=src, group=self.pg)
            dist.broadcast(self.flat_buf, src=src, group=self.pg)
            self.broadcast_buffers(src)

    def broadcast_buffers(self, src=0):
        """BatchNorm statistics stay process-local during training (the reference has no SyncBN); call this before
        saving a checkpoint if every rank must hold rank ``src``'s buffers."""
        if not self.distributed:
            return
        bufs = [b for b in self.model.buffers()]
        for dtype in dict.fromkeys(b.dtype for b in bufs):      # first-seen order: identical on every rank (a set's is not)
            group = [b for b in bufs if b.dtype == dtype]
            flat = torch.cat([b.detach().reshape(-1) for b in group])
            dist.broadcast(flat, src=src, group=self.pg)
            off = 0
            with torch.no_grad():
                for b in group:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()

    # ---- gradient exchange ---------------------------------------------------------------------------------------
    def _make_hook(self, i):
        def hook(param):
            if self._fired is None:            # outside forward_backward (e.g. a user's own backward): nothing to do
                return
            self._fired[i] += 1
            b = self._bucket_of[i]
            if self._expected is None:         # planning step: count, remember when each bucket saw its last accumulation
                self._seq += 1
                self._last_fire[b] = self._seq
                return
            if self._fired[i] > self._expected[i]:
                raise RuntimeError("FlatSGDTrainer: a parameter accumulated a gradient more often than in the planning step -- the "
                                   "autograd graph changed (stage recomputation switched on, a block now shared, a branch newly "
                                   "used).  Call trainer.replan() on every rank before the next step.")
            if b in self._static or self._fired[i] < self._expected[i]:
                return
            self._pending[b] -= 1
            if self._pending[b] == 0:
                self._ready[b] = True
                self.last_completion_order.append(b)       # (diagnostic: THIS rank's completion order, next to the launch order)
                # complete buckets leave strictly in the agreed order, whatever order THIS rank's backward completed them in
                while self._next < len(self._order) and self._ready[self._order[self._next]]:
                    self._launch_bucket(self._order[self._next])
                    self._next += 1
        return hook

    def replan(self):
        """Forget the exchange plan: the next step is a planning step again (collective: call it on every rank)."""
        self._expected, self._order, self._static, self.deferred_buckets = None, None, set(), 0

    def _make_plan(self):
        """After the planning step's backward: agree on the accumulation counts and on the launch order."""
        dev = self.flat_grad.device
        counts = torch.tensor(self._fired, dtype=torch.int32, device=dev)
        hi, lo = counts.clone(), counts.clone()
        dist.all_reduce(hi, op=dist.ReduceOp.MAX, group=self.pg)
        dist.all_reduce(lo, op=dist.ReduceOp.MIN, group=self.pg)
        expected, agreed = hi.tolist(), (hi == lo).tolist()
        static = set()
        for b, bk in enumerate(self.buckets):
            if not all(agreed[i] for i in bk["params"]) or not any(expected[i] for i in bk["params"]):
                static.add(b)
        # rank 0's completion order (buckets it never completed go last); broadcast so that every rank launches in that order
        mine = sorted((b for b in range(len(self.buckets)) if b not in static), key=lambda b: (self._last_fire[b] == 0, self._last_fire[b], b))
        order = torch.tensor(mine + [-1] * (len(self.buckets) - len(mine)), dtype=torch.int32, device=dev)
        dist.broadcast(order, src=dist.get_global_rank(self.pg, 0) if self.pg is not None else 0, group=self.pg)
        self._expected, self._static = expected, static
        self._order = [int(b) for b in order.tolist() if b >= 0]
        self.deferred_buckets = len(static)

    def _pack_bucket(self, b):
        dst, src = [], []
        for i in self.buckets[b]["params"]:
            p, v = self.params[i], self.grad_views[i]
            if p.grad is None:          # parameter not reached by this step's graph (unused / frozen branch)
                v.zero_()
                self._skipped.append(i)
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)

    def _launch_bucket(self, b):
        """Pack bucket b's gradients into their flat range and start its sum all-reduce (asynchronous: RCCL runs it
        on its own stream while backward keeps producing the earlier layers' gradients)."""
        self._pack_bucket(b)
        self._launched[b] = True
        if self.distributed:
            bk = self.buckets[b]
            self._works[b] = dist.all_reduce(self.flat_grad[bk["lo"]:bk["hi"]], op=dist.ReduceOp.SUM, group=self.pg, async_op=True)

    def forward_backward(self, corrupted, mask, clean_nhwc):
        for p in self.params:
            p.grad = None
        self._works, self._skipped = {}, []
        self._launched = [False] * len(self.buckets)
        planning = self.overlap and self._expected is None
        if self.overlap:
            self._fired = [0] * len(self.params)
            if planning:
                self._seq, self._last_fire = 0, [0] * len(self.buckets)
            else:
                # a bucket is complete when every parameter the plan expects a gradient for has accumulated its planned count
                self._pending = [sum(1 for i in bk["params"] if self._expected[i] > 0) for bk in self.buckets]
                self._ready, self._next = [False] * len(self.buckets), 0
        self.last_completion_order = []
        with deferred_batch_counters():      # the BatchNorm batch counters: one multi-tensor add instead of one kernel each
            out = self.model((corrupted, mask))
        loss = self.loss_fn(out, clean_nhwc)
        # mean over ranks folded into the backward seed: sum-all-reduce then yields the average
        loss.backward(torch.full((), 1.0 / self.world, dtype=torch.float32, device=loss.device))
        if planning:
            self._make_plan()
        elif self.overlap:
            short = [i for i in range(len(self.params)) if self._fired[i] < self._expected[i] and self._bucket_of[i] not in self._static]
            if short:
                raise RuntimeError(f"FlatSGDTrainer: {len(short)} parameter(s) accumulated fewer gradients than in the planning step -- the "
                                   "autograd graph changed; call trainer.replan() on every rank before the next step.")
        self._fired = self._pending = None
        return loss.detach()    # do not hand the autograd graph (and its AccumulateGrad nodes) to the caller

    def _pack_gradients(self):
        self._skipped = []
        for b in range(len(self.buckets)):
            self._pack_bucket(b)

    def reduce_gradients(self):
        """Finish the exchange: buckets whose hooks did not fire (no overlap, or parameters without a gradient) are
        packed and reduced now, then every outstanding all-reduce is waited for."""
        launched = self._launched or [False] * len(self.buckets)
        for b in range(len(self.buckets)):
            if not launched[b]:
                self._launch_bucket(b) if self.distributed else self._pack_bucket(b)
        t0 = self._exposed_begin()
        for b in sorted(self._works):
            self._works[b].wait()
        self._exposed_end(t0)
        self._works, self._launched = {}, None
        self._skipped = sorted(set(self._skipped))
        for i, (p, v) in enumerate(zip(self.params, self.grad_views)):
            p.grad = v

    # exposed communication = how long the compute stream stalls in reduce_gradients() for collectives that backward did not
    # hide (HIP events on the current stream around the waits: nothing else is enqueued between them; host clock on CPU / gloo)
    def _exposed_begin(self):
        if not (self.measure_exposed and self.distributed):
            return None
        if self.flat_grad.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            return e
        import time
        return time.perf_counter()

    def _exposed_end(self, t0):
        if t0 is None:
            return
        if self.flat_grad.is_cuda:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._exposed.append((t0, e))
        else:
            import time
            self._exposed.append((time.perf_counter() - t0) * 1e3)

    def exposed_ms(self):
        """Mean stall per step recorded while ``measure_exposed`` was set (synchronises); None when nothing was recorded."""
        if not self._exposed:
            return None
        if self.flat_grad.is_cuda:
            torch.cuda.synchronize(self.flat_grad.device)
            vals = [a.elapsed_time(b) for a, b in self._exposed]
        else:
            vals = list(self._exposed)
        self._exposed = []
        return sum(vals) / len(vals)

    def update(self):
        """Fused SGD-Nesterov over the flat buffers.  Parameters that received no gradient in this step keep their value AND
        their momentum buffer, as with torch.optim.SGD (which skips ``p.grad is None``; the reference's optimizer) -- their
        slices are saved around the fused kernel instead of breaking it up.  Ranks must agree on which parameters those are
        (same graph on every rank, the DistributedDataParallel ``find_unused_parameters=False`` contract)."""
        keep = [(self.slices[i][0], self.slices[i][1]) for i in self._skipped]
        saved = [(self.flat_param[o:o + n].clone(), self.flat_buf[o:o + n].clone()) for o, n in keep]
        ops.sgd_nesterov_(self.flat_param, self.flat_grad, self.flat_buf, self.lr, self.momentum, self.weight_decay)
        for (o, n), (p0, b0) in zip(keep, saved):
            self.flat_param[o:o + n].copy_(p0)
            self.flat_buf[o:o + n].copy_(b0)

    def step(self, corrupted, mask, clean_nhwc):
        loss = self.forward_backward(corrupted, mask, clean_nhwc)
        self.reduce_gradients()
        self.update()
        return loss

    def comm_stats(self, iters=10):
        """What bench.py prints for N > 1: ranks the collective library saw, bucket layout, and the time / bus bandwidth
        of all-reducing the whole gradient buffer bucket by bucket (blocking, after a barrier; ring bus bandwidth =
        2 (n-1)/n x bytes / time)."""
        info = {"world": self.world, "backend": dist.get_backend(self.pg) if self.distributed else None,
                "buckets": len(self.buckets), "bucket_mb": [round((bk["hi"] - bk["lo"]) * 4 / 2**20, 2) for bk in self.buckets],
                "grad_mb": round(self.flat_grad.numel() * 4 / 2**20, 2), "overlap_with_backward": self.overlap,
                "deferred_buckets": self.deferred_buckets}
        if not self.distributed or self.world < 2:
            return info
        import time
        scratch = torch.zeros_like(self.flat_grad)
        sync = torch.cuda.synchronize if scratch.is_cuda else (lambda: None)
        for timed in (False, True):
            dist.barrier(group=self.pg)
            sync()
            t0 = time.perf_counter()
            for _ in range(iters if timed else 2):
                works = [dist.all_reduce(scratch[bk["lo"]:bk["hi"]], group=self.pg, async_op=True) for bk in self.buckets]
                for w in works:
                    w.wait()
            sync()
            dt = (time.perf_counter() - t0) / (iters if timed else 2)
        nbytes = scratch.numel() * 4
        info.update({"allreduce_ms": round(dt * 1e3, 3), "alg_gb_s": round(nbytes / dt / 1e9, 2),
                     "bus_gb_s": round(2 * (self.world - 1) / self.world * nbytes / dt / 1e9, 2)})
        exposed = self.exposed_ms()
        if exposed is not None:
            # exposed: what the step actually waited for; overlapped: the rest of the stand-alone all-reduce time, hidden
            # behind backward
            info.update({"exposed_ms_per_step": round(exposed, 3), "overlapped_ms_per_step": round(max(0.0, dt * 1e3 - exposed), 3)})
        return info

    # ---- HIP-graph replay of the step --------------------------------------------------------------------------
    # One step is ~900 kernel launches from Python; captured once into a HIP graph the whole chain is replayed
    # with a single launch, which removes the host-side gaps between kernels (~2.5 % of the step on MI355X).
    def capture(self, corrupted, mask, clean_nhwc, warmup=2):
        """Capture forward + loss + backward + gradient packing (+ the SGD update when there is no all-reduce)
        for batches of this shape.  The tensors passed here become the graph's static input buffers;
        ``step_graph`` copies each new batch into them.  Parameters, optimizer state and BatchNorm buffers
        are restored after the warm-up runs, so capturing does not advance training."""
        dev = self.flat_param.device
        self._static_in = (corrupted, mask, clean_nhwc)
        self.overlap = False           # a captured step cannot launch collectives from hooks: exchange after the replay
        buffers = [b for b in self.model.buffers()]
        saved = [self.flat_param.clone(), self.flat_buf.clone()] + [b.clone() for b in buffers]
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            for _ in range(max(1, warmup)):
                self.forward_backward(corrupted, mask, clean_nhwc)
                self._pack_gradients()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)

        def restore():
            with torch.no_grad():
                self.flat_param.copy_(saved[0])
                self.flat_buf.copy_(saved[1])
                for b, v in zip(buffers, saved[2:]):
                    b.copy_(v)

        restore()
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._static_loss = self.forward_backward(corrupted, mask, clean_nhwc)
            self._pack_gradients()
            if not self.distributed:
                self.update()
        torch.cuda.synchronize(dev)
        restore()                      # capture does not execute, but keep the contract explicit
        for p, v in zip(self.params, self.grad_views):
            p.grad = v
        return self

    def step_graph(self, corrupted=None, mask=None, clean_nhwc=None):
        """Replay the captured step; new batches are copied into the static input buffers first."""
        for new, static in zip((corrupted, mask, clean_nhwc), self._static_in):
            if new is not None and new.data_ptr() != static.data_ptr():
                static.copy_(new)
        self._graph.replay()
        if self.distributed:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
            self.update()
        return self._static_loss.clone()   # the static tensor is overwritten by the next replay

"""Single-node data-parallel training step for the inpainting nets.

One process per GPU.  Images of a minibatch are independent except for BatchNorm batch
statistics, which the reference computes per process-local batch (no SyncBN), so the batch is
sharded across ranks with replicated weights and ONE exchange per iteration: a sum all-reduce of
the trainable gradients (RCCL over xGMI through ``torch.distributed``, backend "nccl"), packed in
a single flat fp32 buffer (ImageFill: 6.5 M grads = 26 MB -- far below one xGMI link-second, so
one large collective beats bucketing here).  The mean over ranks is folded into the loss scale.
Frozen parameters (``mask_conv.weight``) never enter the buffer.  The update is the SGD-Nesterov
the reference trained with (checkpoints/ReadME.md:4) as one fused HIP kernel over flat buffers.
"""
import torch
import torch.distributed as dist

from . import ops
from .BaseModels import deferred_batch_counters, to_nhwc


class FlatSGDTrainer:
    def __init__(self, model, lr=0.01, momentum=0.9, weight_decay=1e-4, process_group=None, loss_fn=None):
        self.model = model
        self.params = [p for p in model.parameters() if p.requires_grad]
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
        for dtype in {b.dtype for b in bufs}:
            group = [b for b in bufs if b.dtype == dtype]
            flat = torch.cat([b.detach().reshape(-1) for b in group])
            dist.broadcast(flat, src=src, group=self.pg)
            off = 0
            with torch.no_grad():
                for b in group:
                    b.copy_(flat[off:off + b.numel()].view_as(b))
                    off += b.numel()

    def forward_backward(self, corrupted, mask, clean_nhwc):
        for p in self.params:
            p.grad = None
        with deferred_batch_counters():      # the BatchNorm batch counters: one multi-tensor add instead of one kernel each
            out = self.model((corrupted, mask))
        loss = self.loss_fn(out, clean_nhwc)
        # mean over ranks folded into the backward seed: sum-all-reduce then yields the average
        loss.backward(torch.full((), 1.0 / self.world, dtype=torch.float32, device=loss.device))
        return loss.detach()    # do not hand the autograd graph (and its AccumulateGrad nodes) to the caller

    def _pack_gradients(self):
        dst, src = [], []
        for v, p in zip(self.grad_views, self.params):
            if p.grad is None:          # parameter not reached by this step's graph (unused / frozen branch)
                v.zero_()
            elif p.grad.data_ptr() != v.data_ptr():
                dst.append(v)
                src.append(p.grad)
        if dst:
            torch._foreach_copy_(dst, src)

    def reduce_gradients(self):
        self._pack_gradients()
        if self.distributed:
            dist.all_reduce(self.flat_grad, op=dist.ReduceOp.SUM, group=self.pg)
        for p, v in zip(self.params, self.grad_views):
            p.grad = v

    def update(self):
        ops.sgd_nesterov_(self.flat_param, self.flat_grad, self.flat_buf, self.lr, self.momentum, self.weight_decay)

    def step(self, corrupted, mask, clean_nhwc):
        loss = self.forward_backward(corrupted, mask, clean_nhwc)
        self.reduce_gradients()
        self.update()
        return loss

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

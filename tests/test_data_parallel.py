"""N>1 path on CPU: world_size-2 gloo, kernels through the test-only emulator.

The data-parallel step shards the batch, keeps BatchNorm statistics process-local (the reference has no SyncBN), and
exchanges gradients bucket by bucket from ``post_accumulate_grad`` hooks while backward is still running.  Oracle: the
same two half-batches run one after the other in ONE process (fresh replica each), gradients averaged by hand."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _make_model(seed=21, variant="plain", rank=0):
    """variant "reordered": modules registered in the opposite order of their use, so gradients become ready in the reverse of
    the flat layout and the LAST bucket completes first; "shared": one block applied in two recomputed segments, so its
    parameters are accumulated into twice per backward (two post_accumulate_grad firings); "divergent": two parallel branches
    that rank 1 builds in the opposite order AND one of which only rank 1 recomputes in backward -- the same function and the same
    parameters on both ranks, but their backward passes complete the gradient buckets in different orders."""
    import text_segmentation_image_inpainting_amd as T
    from text_segmentation_image_inpainting_amd.memory import checkpoint_segment
    from text_segmentation_image_inpainting_amd.BaseModels import DSConvBlock
    from oracle.filler import fill_state_dict_
    act = nn.LeakyReLU(0.3)

    class Reordered(nn.Module):
        def __init__(self):
            super().__init__()
            self.unused = nn.Parameter(torch.ones(5))
            self.body = T.PartialInvertedResidual(8, 8, 3, 1, 1, 1, 2, BN=True, activation=act, use_1_conv=True, same_holes=True)
            self.stem = T.partial_convolution_block(3, 8, 3, 1, 1, 1, bias=True, BN=False, activation=act)

        def forward(self, args):
            return self.body(self.stem(args))[0]

    class Shared(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = T.partial_convolution_block(3, 8, 3, 1, 1, 1, bias=True, BN=False, activation=act)
            self.block = DSConvBlock(8, 8, 3, 1, 1, BN=True, activation_dep=act, activation_point=act)
            self.tail = DSConvBlock(8, 8, 3, 1, 1, BN=True, activation_dep=act, activation_point=None)
            self.unused = nn.Parameter(torch.ones(5))

        def forward(self, args):
            h = self.stem(args)[0]
            h = checkpoint_segment(self.block, h)
            h = checkpoint_segment(self.block, h)
            return self.tail(h)

    class Divergent(nn.Module):
        def __init__(self, flip):
            super().__init__()
            self.flip = flip
            self.stem = T.partial_convolution_block(3, 8, 3, 1, 1, 1, bias=True, BN=False, activation=act)
            self.left = DSConvBlock(8, 8, 3, 1, 1, BN=True, activation_dep=act, activation_point=act)
            self.right = DSConvBlock(8, 8, 3, 1, 1, BN=True, activation_dep=act, activation_point=act)
            self.tail = DSConvBlock(8, 8, 3, 1, 1, BN=True, activation_dep=act, activation_point=None)
            self.unused = nn.Parameter(torch.ones(5))

        def forward(self, args):
            from text_segmentation_image_inpainting_amd import ops
            from text_segmentation_image_inpainting_amd.BaseModels import to_nchw, to_nhwc
            h = self.stem(args)[0]
            if self.flip:       # rank 1: right first, and recomputed in backward (its gradients arrive when the recomputation runs)
                r = checkpoint_segment(self.right, h)
                l = self.left(h)
            else:
                l = self.left(h)
                r = self.right(h)
            return self.tail(to_nchw(ops.add_act(to_nhwc(l), to_nhwc(r))))

    if variant == "divergent":
        net = Divergent(flip=rank == 1)
        fill_state_dict_(net.state_dict(), seed=seed)
        return net
    if variant != "plain":
        net = {"reordered": Reordered, "shared": Shared}[variant]()
        fill_state_dict_(net.state_dict(), seed=seed)
        return net

    class Net(nn.Module):
        """stem partial conv + a PartialInvertedResidual with three BatchNorms + a head that ignores one parameter"""

        def __init__(self):
            super().__init__()
            self.stem = T.partial_convolution_block(3, 8, 3, 1, 1, 1, bias=True, BN=False, activation=act)
            self.body = T.PartialInvertedResidual(8, 8, 3, 1, 1, 1, 2, BN=True, activation=act, use_1_conv=True, same_holes=True)
            self.unused = nn.Parameter(torch.ones(5))          # never reaches the loss: its gradient stays None

        def forward(self, args):
            return self.body(self.stem(args))[0]
    net = Net()
    fill_state_dict_(net.state_dict(), seed=seed)
    return net


def _data():
    from oracle.filler import seeded_input
    x, mask = seeded_input(4, 3, 10, 10, seed=21, hole_frac=0.2, per_channel_mask=True)
    tgt = torch.from_numpy(np.random.default_rng(5).standard_normal((4, 8, 10, 10)).astype(np.float32))
    return x, mask, tgt


def _trainer(model, overlap=True):
    from text_segmentation_image_inpainting_amd.train_step import FlatSGDTrainer
    return FlatSGDTrainer(model, lr=0.1, momentum=0.9, weight_decay=1e-3, bucket_mb=0.0005, overlap=overlap)   # ~130 floats: several buckets


def _worker(rank, world, initfile, out, variant="plain", overlap=True):
    from tests.backends import emu_backend
    from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    with emu_backend():
        x, mask, tgt = _data()
        sl = slice(rank * 2, rank * 2 + 2)
        model = _make_model(seed=21 if rank == 0 else 99, variant=variant, rank=rank)     # rank 1 starts from different weights AND buffers ...
        if rank == 1:
            for b in model.buffers():
                if b.dtype.is_floating_point:
                    b.add_(0.5)
        tr = _trainer(model, overlap)
        assert tr.overlap == overlap and len(tr.buckets) >= 3
        tr.broadcast_parameters()                                  # ... until rank 0's are broadcast
        start_bufs = {k: v.clone() for k, v in model.state_dict().items() if "running" in k}
        unused0 = model.unused.detach().clone()
        order, ready = [], []                                      # the order in which buckets went out / became complete
        launch = tr._launch_bucket
        tr._launch_bucket = lambda b: (order.append(b), launch(b))[1]
        tr.measure_exposed = True
        losses = []
        for _ in range(3):
            losses.append(float(tr.step(x[sl], mask[sl], to_nhwc(tgt[sl]))))
            ready.append(list(tr.last_completion_order))
        stats = tr.comm_stats(iters=2)
    torch.save({"p": tr.flat_param.clone(), "g": tr.flat_grad.clone(), "loss": losses, "start_bufs": start_bufs,
                "bufs": {k: v.clone() for k, v in model.state_dict().items() if "running" in k or "tracked" in k},
                "stats": stats, "order": order, "plan": (tr._order, sorted(tr._static), tr._expected), "ready": ready, "unused": model.unused.detach().clone(), "unused0": unused0,
                "hooks": len(tr._hooks)}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("variant,overlap", [("plain", True), ("plain", False), ("reordered", True), ("shared", True), ("divergent", True)])
def test_dp2_bucketed_overlap_matches_sequential_shards(variant, overlap):
    """plain: buckets complete in layout order and go out from the hooks; overlap=False: everything is reduced after
    backward; reordered: buckets complete out of layout order; shared: a block whose parameters accumulate twice per backward --
    its buckets go out after the SECOND accumulation; divergent: the two ranks' backward passes complete the buckets in DIFFERENT
    orders (branches built in opposite order, one recomputed on rank 1 only) and still issue their collectives in one agreed order.
    Step 1 is the planning step (exchange after backward), steps 2 and 3 overlap.  All five must equal the sequential oracle."""
    from tests.backends import emu_backend
    from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
    from text_segmentation_image_inpainting_amd import ops
    with tempfile.TemporaryDirectory() as d:
        initfile, out = os.path.join(d, "init"), os.path.join(d, "out")
        mp.spawn(_worker, args=(2, initfile, out, variant, overlap), nprocs=2, join=True)
        r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    nb = r0["stats"]["buckets"]
    # every bucket exactly once per step, and the SAME sequence of collectives on both ranks
    assert r0["order"] == r1["order"] and len(r0["order"]) == 3 * nb
    steps = [r0["order"][s * nb:(s + 1) * nb] for s in range(3)]
    assert all(sorted(s) == list(range(nb)) for s in steps)
    assert steps[0] == list(range(nb))                             # planning step (or no overlap): after backward, in layout order
    assert steps[1] == steps[2]
    if overlap:
        plan_order, static, expected = r0["plan"]
        assert r0["plan"] == r1["plan"]
        assert steps[1] == plan_order + static                     # the agreed order first, what never overlaps after backward
        assert r0["stats"]["deferred_buckets"] == len(static) <= 1     # at most the bucket that holds nothing but the unused parameter
    if variant == "plain":
        assert steps[1] == list(range(nb)) and r0["stats"]["overlap_with_backward"] == overlap
    if variant == "reordered":
        assert steps[1] != list(range(nb)) and steps[1][0] != 0    # a later bucket went out before bucket 0
    if variant == "shared":
        assert max(expected) == 2                                  # the shared block's parameters: two accumulations per backward
    assert r0["hooks"] == (r0["stats"]["world"] > 1 and overlap) * sum(1 for _ in _make_model(variant=variant).parameters() if _.requires_grad)
    if variant == "divergent":
        assert r0["ready"] != r1["ready"], "the variant is meant to make the two ranks complete their buckets in different orders"
    assert r0["stats"]["exposed_ms_per_step"] >= 0 and "overlapped_ms_per_step" in r0["stats"]
    assert torch.equal(r0["unused"], r0["unused0"])                # no gradient -> no weight decay, no momentum (torch.optim.SGD)
    assert torch.equal(r0["p"], r1["p"]) and torch.equal(r0["g"], r1["g"])      # replicas stay in lock-step
    assert all(torch.equal(r0["start_bufs"][k], r1["start_bufs"][k]) for k in r0["start_bufs"])   # buffers were broadcast
    assert r0["stats"]["world"] == 2 and r0["stats"]["buckets"] >= 3 and r0["stats"]["allreduce_ms"] > 0
    # oracle: one process, the two shards one after the other on replicas of the broadcast state
    with emu_backend():
        x, mask, tgt = _data()
        reps = [_trainer(_make_model(seed=21, variant=variant, rank=r)) for r in range(2)]
        for step in range(3):
            grads = []
            for r, tr in enumerate(reps):
                sl = slice(r * 2, r * 2 + 2)
                tr.world = 2                                    # same 1/world loss scale as the ranks used
                tr.forward_backward(x[sl], mask[sl], to_nhwc(tgt[sl]))
                tr._pack_gradients()
                grads.append(tr.flat_grad.clone())
            mean = grads[0] + grads[1]
            for tr in reps:
                tr.flat_grad.copy_(mean)
                tr.update()
        ref = reps[0]
    assert torch.allclose(r0["g"], ref.flat_grad, rtol=1e-5, atol=1e-7)
    assert torch.allclose(r0["p"], ref.flat_param, rtol=1e-5, atol=1e-7)
    # BatchNorm statistics are per rank (no SyncBN): rank r's buffers = replica r's, and they differ between ranks
    for r, rec in enumerate((r0, r1)):
        sd = reps[r].model.state_dict()
        for k, v in rec["bufs"].items():
            assert torch.allclose(v.float(), sd[k].float(), rtol=1e-5, atol=1e-6), (r, k)
    assert any(not torch.equal(r0["bufs"][k], r1["bufs"][k]) for k in r0["bufs"] if "running_mean" in k)
    # batch counters: one per application and step -- the shared block runs twice per step, and its recomputation in backward
    # does not count again
    assert int(r0["bufs"][next(k for k in r0["bufs"] if "tracked" in k)]) == (6 if variant == "shared" else 3)


@pytest.mark.gpu
def test_dp2_torchrun_rccl_gpu():
    """The same step under ``torchrun --nproc-per-node 2`` with the "nccl" (= RCCL) backend; skips itself on a
    single-GPU box."""
    import subprocess
    import sys
    if torch.cuda.device_count() < 2:
        pytest.skip("needs 2 GPUs")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "2",
           "--size", "128", "--no-cpu-baseline", "--no-f32-leg"]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["comm"]["world"] == 2 and line["comm"]["backend"] == "nccl"

"""N>1 path on CPU: world_size-2 gloo, kernels through the test-only emulator.  Checks that the
sharded step (per-rank half batch, flat-gradient sum all-reduce, fused SGD) equals the
single-process step on the whole batch."""
import os
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _make_model():
    import text_segmentation_image_inpainting_amd as T
    from oracle.filler import fill_state_dict_
    m = nn.Sequential(T.partial_convolution_block(3, 4, 3, 1, 1, 1, bias=True, BN=False, activation=nn.LeakyReLU(0.3)))

    class Net(nn.Module):
        def __init__(self):
            super().__init__()
            self.body = m

        def forward(self, args):
            return self.body[0](args)[0]
    net = Net()
    fill_state_dict_(net.state_dict(), seed=21)
    return net


def _data():
    from oracle.filler import seeded_input
    x, mask = seeded_input(4, 3, 10, 10, seed=21, hole_frac=0.2, per_channel_mask=True)
    tgt = torch.from_numpy(np.random.default_rng(5).standard_normal((4, 4, 10, 10)).astype(np.float32))
    return x, mask, tgt


def _one_step(x, mask, tgt):
    from text_segmentation_image_inpainting_amd.BaseModels import to_nhwc
    from text_segmentation_image_inpainting_amd.train_step import FlatSGDTrainer
    tr = FlatSGDTrainer(_make_model(), lr=0.1, momentum=0.9, weight_decay=1e-3)
    tr.broadcast_parameters()
    loss = tr.step(x, mask, to_nhwc(tgt))
    return tr.flat_param.clone(), tr.flat_grad.clone(), float(loss)


def _worker(rank, world, initfile, out):
    from tests.backends import emu_backend
    dist.init_process_group("gloo", init_method="file://" + initfile, rank=rank, world_size=world)
    with emu_backend():
        x, mask, tgt = _data()
        sl = slice(rank * 2, rank * 2 + 2)
        p, g, loss = _one_step(x[sl], mask[sl], tgt[sl])
    torch.save({"p": p, "g": g, "loss": loss}, f"{out}.{rank}")
    dist.barrier()
    dist.destroy_process_group()


def test_dp2_matches_single_process():
    from tests.backends import emu_backend
    with emu_backend():
        x, mask, tgt = _data()
        p_ref, g_ref, loss_ref = _one_step(x, mask, tgt)
    with tempfile.TemporaryDirectory() as d:
        initfile, out = os.path.join(d, "init"), os.path.join(d, "out")
        mp.spawn(_worker, args=(2, initfile, out), nprocs=2, join=True)
        r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert torch.equal(r0["p"], r1["p"]) and torch.equal(r0["g"], r1["g"])      # ranks stay in lock-step
    assert torch.allclose(r0["g"], g_ref, rtol=1e-5, atol=1e-7)                  # mean of shard grads = full-batch grad
    assert torch.allclose(r0["p"], p_ref, rtol=1e-5, atol=1e-7)
    assert abs(0.5 * (r0["loss"] + r1["loss"]) - loss_ref) < 1e-6

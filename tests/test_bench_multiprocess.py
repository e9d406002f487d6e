"""bench.py's N > 1 path on the CPU container: two processes rendezvous like under ``torch.distributed.run`` (RANK / WORLD_SIZE /
MASTER_* in the environment), run the timed loop, the stand-alone all-reduce measurement (``comm_stats``), the exposed /
overlapped split, the max-over-ranks clock and the JSON line -- with ``gloo`` and the kernel emulator standing in for RCCL and the
GPU (bench.TEST_RUNTIME).  What it pins: the first real 2 / 4 / 8-GPU run cannot die in launch or reporting code that has never
executed with world > 1."""
import json
import os
import socket
import tempfile

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _rank(rank, world, port, out, bodies=1, bucket_mb=0.0005):
    import contextlib
    import io
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import bench
    from tests.backends import emu_backend
    import text_segmentation_image_inpainting_amd as T
    act = torch.nn.LeakyReLU(0.3)

    class Tiny(torch.nn.Module):
        """3 -> 8 stem + one inverted residual + a 3-channel head: every kernel family of a step, seconds on the emulator"""

        def __init__(self):
            super().__init__()
            self.stem = T.partial_convolution_block(3, 8, 3, 1, 1, 1, bias=True, BN=False, activation=act)
            self.body = torch.nn.Sequential(*[T.PartialInvertedResidual(8, 8, 3, 1, 1, 1, 2, BN=True, activation=act, use_1_conv=True, same_holes=True)
                                              for _ in range(bodies)])
            self.head = T.partial_convolution_block(8, 3, 3, 1, 1, 1, bias=True, BN=False, activation=False)

        def forward(self, args):
            h = self.stem(args)
            for blk in self.body:
                h = blk(h)
            return self.head(h)[0]
    with emu_backend() as dev:
        bench.TEST_RUNTIME = {"device": dev, "backend": "gloo", "model_factory": Tiny, "trainer_kwargs": {"bucket_mb": bucket_mb}}
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main(["--gpus", str(world), "--steps", "2", "--warmup", "1", "--batch", "2", "--size", "32", "--no-cpu-baseline", "--no-f32-leg"])
    open(f"{out}.{rank}", "w").write(buf.getvalue())


def _run(world, bodies=1, bucket_mb=0.0005):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "out")
        mp.spawn(_rank, args=(world, _free_port(), out, bodies, bucket_mb), nprocs=world, join=True)
        texts = [open(f"{out}.{r}").read() for r in range(world)]
    lines = [ln for ln in texts[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not any(ln.startswith("{") for t in texts[1:] for ln in t.splitlines()), "exactly one JSON line, from rank 0"
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == world and rec["steps"] == 2 and rec["warmup"] == 1 and rec["scaling"] == "weak" and rec["higher_is_better"] is True
    assert rec["config"]["global_batch"] == 2 * world and rec["config"]["parallelism"] == f"dp{world}"
    # whole-job images / max-over-ranks step time
    assert rec["value"] > 0 and abs(rec["value"] - 2 * world / (rec["ms_per_step"] / 1e3)) <= 0.05 * rec["value"] + 0.01
    comm = rec["comm"]
    assert comm["world"] == world and comm["backend"] == "gloo" and comm["buckets"] >= 2 and comm["allreduce_ms"] > 0
    assert comm["exposed_ms_per_step"] >= 0 and comm["overlapped_ms_per_step"] >= 0 and comm["overlap_with_backward"] is True
    assert rec["forward_only"]["value"] > 0 and rec["cpu_baseline"] is None if "cpu_baseline" in rec else True
    return rec


def test_bench_two_ranks_gloo():
    _run(2)


def test_bench_four_ranks_gloo():
    """the 4-rank rung of the 1 / 2 / 4 / 8 ladder the driver runs: rendezvous, bucketed all-reduce, max-over-ranks clock, one line"""
    _run(4)


def test_bench_two_ranks_many_buckets_gloo():
    """a bucket layout like ImageFillOrigin's (131 MB of gradients = 9 buckets of 16 MB): here 3 blocks cut into >= 9 buckets, so the
    hooks launch, complete and wait on many asynchronous all-reduces per step, in gradient-ready order"""
    rec = _run(2, bodies=3, bucket_mb=0.0004)
    assert rec["comm"]["buckets"] >= 9, rec["comm"]


def _rank_seg(rank, world, port, out):
    """the segmentation branch of bench.py (BinaryFocalLoss step) in bf16 ACTIVATION STORAGE under two gloo ranks"""
    import contextlib
    import io
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    import bench
    from tests.backends import emu_backend
    from text_segmentation_image_inpainting_amd.BaseModels import Conv2d, ConvSpec, build_chain, run_chain

    class TinySeg(torch.nn.Module):
        """fp32 image -> stride-2 stem (space-to-depth entry of the bf16 path) -> 3x3 conv + BatchNorm -> 1-channel logits (fp32) -> x2"""

        def __init__(self):
            super().__init__()
            act = torch.nn.LeakyReLU(0.3)
            self.body = torch.nn.Sequential(*build_chain(3, (ConvSpec(32, 3, 2, 1), ConvSpec(16, 3, 1, 1)), act)[0], Conv2d(16, 1, 3, 1, 1))

        def forward(self, x):
            return torch.nn.functional.interpolate(run_chain(list(self.body), x), scale_factor=2, mode="nearest")
    with emu_backend() as dev:
        bench.TEST_RUNTIME = {"device": dev, "backend": "gloo", "model_factory": TinySeg}
        buf = io.StringIO()
        with contextlib.redirect_stdout(buf):
            bench.main(["--gpus", str(world), "--model", "XceptionTextSegment", "--storage", "bf16", "--steps", "2", "--warmup", "1", "--batch", "2",
                        "--size", "32", "--no-cpu-baseline"])
    open(f"{out}.{rank}", "w").write(buf.getvalue())


def test_bench_two_ranks_gloo_bf16_storage_segmentation_step():
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "out")
        mp.spawn(_rank_seg, args=(2, _free_port(), out), nprocs=2, join=True)
        texts = [open(f"{out}.{r}").read() for r in range(2)]
    lines = [ln for ln in texts[0].splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and not any(ln.startswith("{") for ln in texts[1].splitlines())
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["config"]["activation_storage"] == "bf16" and rec["dtype"].startswith("bf16 activation")
    assert rec["config"]["global_batch"] == 4 and rec["value"] > 0 and rec["comm"]["world"] == 2 and rec["comm"]["allreduce_ms"] > 0
    assert rec["final_loss"] == rec["final_loss"] and rec["final_loss"] > 0          # finite

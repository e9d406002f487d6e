"""Two ways to run the SAME parity checks:

* ``gpu``  -- the product: libtsii_hip.so on an MI355X (tests marked ``@pytest.mark.gpu``).
* ``emu``  -- TEST-ONLY: the unmodified kernel sources compiled against tests/emu (host SIMT
  emulation) so index math / tiling / MFMA fragment layouts are checked in the CPU container.
  It is wired in by monkeypatching the loader from the test process; the product package has
  no knowledge of it and no CPU path of its own.
"""
import contextlib
import ctypes

import pytest
import torch

from text_segmentation_image_inpainting_amd import _lib


@contextlib.contextmanager
def emu_backend():
    from tests.emu import build_emu
    if not build_emu.available():
        pytest.skip("host clang++ for the emulator build is not available")
    from text_segmentation_image_inpainting_amd import ops
    cdll = _lib.bind(ctypes.CDLL(build_emu.build()))
    saved = (_lib._LIB, _lib.stream, _lib.check_device, ops._ws)
    # every kernel workspace the package allocates gets a canary tail right behind the size the library asked for, checked on the
    # way out: on the host an overrun lands in the heap silently
    canary, guarded = -12345.678, []

    def guarded_ws(nbytes, like):
        n = max(4, (int(nbytes) + 3) // 4)
        buf = torch.empty(n + 64, dtype=torch.float32)
        buf[n:] = canary
        guarded.append((buf, n))
        return buf[:n]
    # host tensors stand in for device tensors; the dtype rule of check_device stays in force
    _lib._LIB, _lib.stream, _lib.check_device, ops._ws = cdll, (lambda: None), _lib.check_dtype, guarded_ws
    try:
        yield torch.device("cpu")
        for buf, n in guarded:
            assert bool((buf[n:] == canary).all()), f"a kernel wrote past its {4 * n}-byte workspace"
    finally:
        _lib._LIB, _lib.stream, _lib.check_device, ops._ws = saved


@contextlib.contextmanager
def gpu_backend():
    if not torch.cuda.is_available():
        pytest.fail("-m gpu tests need a ROCm GPU")
    _lib.lib()  # fails loudly if libtsii_hip.so is missing
    yield torch.device("cuda:0")
    torch.cuda.synchronize()


BACKENDS = {"emu": emu_backend, "gpu": gpu_backend}


def both_backends(fn):
    """Decorator: generate test_<name>[emu] (CPU suite) and test_<name>[gpu] (-m gpu)."""
    return pytest.mark.parametrize(
        "backend", [pytest.param("emu"), pytest.param("gpu", marks=pytest.mark.gpu)])(fn)

#!/usr/bin/env python3
"""Samples the GPU's engine clock / power (rocm-smi) while a GEMM shape runs in a loop: is the matrix pipe running at the nominal
2.4 GHz under this load, or is the chip power-managed below it?
    python tools/clock_watch.py M K N seconds"""
import os
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402


def main():
    M, K, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
    secs = float(sys.argv[4]) if len(sys.argv) > 4 else 3.0
    from text_segmentation_image_inpainting_amd import _lib
    from text_segmentation_image_inpainting_amd._lib import call, ptr
    L = _lib.lib()
    dev = torch.device("cuda:0")
    st = _lib.stream()
    x = torch.randn(M, K, device=dev)
    w = torch.randn(N, K, device=dev) * 0.05
    y = torch.empty(M, N, device=dev)
    wws = torch.empty(L.tsii_pw_ws_bytes(N, K) // 4 + 4, device=dev)
    fn = lambda: call("tsii_pw_fwd", ptr(x), M, K, ptr(w), N, None, None, 0, None, None, None, ptr(y), ptr(wws), wws.numel() * 4, st)
    samples, stop = [], False

    def watch():
        while not stop:
            r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True)
            samples.append(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr.strip()[-200:])
            time.sleep(0.2)
    th = threading.Thread(target=watch)
    th.start()
    t0 = time.perf_counter()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    while time.perf_counter() - t0 < secs:
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        n += 1
        last = e0.elapsed_time(e1) / 50
    stop = True
    th.join()
    hdr = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], capture_output=True, text=True).stdout.strip().splitlines()
    print("ms per launch (last batch): %.3f" % last)
    print(hdr[0] if hdr else "")
    for s in samples[:: max(1, len(samples) // 8)]:
        print(s)


if __name__ == "__main__":
    main()

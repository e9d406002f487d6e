"""TEST-ONLY: compile the unmodified csrc/*.hip with the host clang++ against the HIP
emulation header in tests/emu/hip/ -> tests/emu/_build/libtsii_emu.so (see that header)."""
import hashlib
import os
import subprocess
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
CSRC = os.path.join(ROOT, "text_segmentation_image_inpainting_amd", "csrc")
OUT = os.path.join(HERE, "_build")
LIB = os.path.join(OUT, "libtsii_emu.so")
CLANG = "/opt/rocm/lib/llvm/bin/clang++"
FLAGS = ["-x", "c++", "-std=c++17", "-O2", "-fPIC", "-I", HERE, "-Wno-psabi", "-Wno-unknown-pragmas",
         "-Wno-unknown-attributes", "-Wno-ignored-attributes", "-Wno-pass-failed"]


def available():
    return os.path.exists(CLANG)


def _one(job):
    src, obj, dig = job
    stamp = obj + ".sha"
    if os.path.exists(obj) and os.path.exists(stamp) and open(stamp).read() == dig:
        return src, 0, ""
    r = subprocess.run([CLANG] + FLAGS + ["-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode == 0:
        open(stamp, "w").write(dig)
    return src, r.returncode, r.stdout + r.stderr


def build():
    os.makedirs(OUT, exist_ok=True)
    hdr = hashlib.sha256()
    for d, names in ((CSRC, os.listdir(CSRC)), (os.path.join(HERE, "hip"), os.listdir(os.path.join(HERE, "hip")))):
        for n in sorted(names):
            if n.endswith(".h"):
                hdr.update(open(os.path.join(d, n), "rb").read())
    hdr.update(open(os.path.join(ROOT, "include", "tsii_hip.h"), "rb").read())
    srcs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    srcs.append(os.path.join(HERE, "emu_runtime.cpp"))
    jobs = []
    for s in srcs:
        h = hdr.copy()
        h.update(open(s, "rb").read())
        jobs.append((s, os.path.join(OUT, os.path.basename(s) + ".o"), h.hexdigest()))
    with ThreadPoolExecutor(8) as ex:
        res = list(ex.map(_one, jobs))
    bad = [(s, o) for s, rc, o in res if rc]
    if bad:
        raise RuntimeError("emu build failed:\n" + "\n".join(o for _, o in bad))
    objs = [j[1] for j in jobs]
    if not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs):
        r = subprocess.run([CLANG, "-shared", "-fPIC", "-o", LIB] + objs, capture_output=True, text=True)
        if r.returncode:
            raise RuntimeError("emu link failed:\n" + r.stdout + r.stderr)
    return LIB


if __name__ == "__main__":
    print(build())

#!/usr/bin/env python3
"""A/B builds of ONE kernel file: recompiles csrc/<file>.hip with extra -D flags and links it with the stock objects into
tools/variants/_bin/libtsii_<name>.so (git-ignored, shipped to the GPU box with the snapshot).  Select a build at run time with
TSII_LIBRARY=<path> (see _lib.py).  Usage: build_variant.py <name> <file.hip> [-DFOO=1 ...]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from text_segmentation_image_inpainting_amd import build_ext as B

def main():
    name, fname, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    B.build(verbose=False)
    out = os.path.join(ROOT, "tools", "variants", "_bin"); os.makedirs(out, exist_ok=True)
    src = os.path.join(B.CSRC, fname)
    obj = os.path.join(out, f"{name}_{os.path.splitext(fname)[0]}.o")
    r = subprocess.run([B._hipcc()] + B.flags_for(src) + extra + ["-c", src, "-o", obj], capture_output=True, text=True)
    if r.returncode:
        sys.exit(r.stdout + r.stderr)
    objs = [os.path.join(B.OBJ, os.path.splitext(os.path.basename(s))[0] + ".o") for s in B.sources() if os.path.basename(s) != fname]
    lib = os.path.join(out, f"libtsii_{name}.so")
    subprocess.check_call([B._hipcc(), "-shared", "-fPIC", "--offload-arch=" + B.ARCH, "-o", lib, obj] + objs)
    os.remove(obj)
    print(lib)

if __name__ == "__main__":
    main()

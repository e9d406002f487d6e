"""Builds csrc/*.hip into libtsii_hip.so for gfx950 with hipcc (in-tree, no JIT cache).

    python -m text_segmentation_image_inpainting_amd.build_ext [--force]

hipcc cross-compiles without a GPU, so this also runs in the CPU-only build container;
the resulting .so travels to the GPU box with the repo snapshot.
"""
import hashlib
import json
import os
import re
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_obj")
LIB = os.path.join(HERE, "libtsii_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", "--offload-arch=" + ARCH, "-Wall", "-Wno-unused-function"]
# every compile also records what each kernel was allocated (VGPRs / AGPRs / scratch / occupancy / LDS) next to its object
# (<obj>.res.json): tests/test_abi_and_host.py holds the scratch policy against it -- no kernel outside a short, measured list spills
RESOURCE_REMARKS = ["-Rpass-analysis=kernel-resource-usage"]
# The MFMA kernels are built without v_pk_{add,mul,fma}_f32 (-packed-fp32-ops): next to a busy matrix pipe the packed forms issue
# at ~1/15 of the scalar rate (measured on MI355X, tools/probes/valu_rates.hip: 9 vs 100-180 cycles per instruction beside a
# v_mfma_f32_32x32x16_bf16 stream).  The stencil / streaming kernels keep them: without MFMAs around they are two flops per slot.
NO_PACKED_F32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
MFMA_SOURCES = ("gemm.hip", "gemm_split.hip", "gemm_pc.hip", "head_mfma.hip", "bf16_gemm.hip")


def flags_for(src):
    return FLAGS + (NO_PACKED_F32 if os.path.basename(src) in MFMA_SOURCES else [])


def _hipcc():
    for cand in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (need ROCm to build libtsii_hip.so)")


_REMARK = re.compile(r"remark: +([A-Za-z ]+?)(?: \[[^\]]*\])?: +(\S+)")
_REMARK_CONT = re.compile(r"^\s+\d+ \| |^\s+\| +\^")


def parse_resource_remarks(lines):
    """{mangled kernel name: {"vgprs", "agprs", "scratch", "occupancy", "lds", "sgprs"}} from -Rpass-analysis=kernel-resource-usage"""
    keys = {"VGPRs": "vgprs", "AGPRs": "agprs", "ScratchSize": "scratch", "Occupancy": "occupancy", "LDS Size": "lds", "TotalSGPRs": "sgprs",
            "VGPRs Spill": "vgpr_spill", "SGPRs Spill": "sgpr_spill"}
    out, cur = {}, None
    for ln in lines:
        m = _REMARK.search(ln)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = out.setdefault(v, {})
        elif cur is not None and k in keys:
            try:
                cur[keys[k]] = int(v)
            except ValueError:
                pass
    return out


def kernel_resources():
    """Resource records of every kernel of the library as built (builds if needed): {source file: {mangled name: record}}"""
    build(verbose=False)
    res = {}
    for src in sources():
        base = os.path.splitext(os.path.basename(src))[0]
        with open(os.path.join(OBJ, base + ".o.res.json")) as f:
            res[os.path.basename(src)] = json.load(f)
    return res


def sources():
    return sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))


def _digest(paths):
    h = hashlib.sha256()
    for p in paths:
        with open(p, "rb") as f:
            h.update(f.read())
    h.update(" ".join(flags_for(paths[0])).encode())
    return h.hexdigest()


def _compile_one(args):
    hipcc, src, obj, stamp, dig = args
    if os.path.exists(obj) and os.path.exists(stamp) and os.path.exists(obj + ".res.json") and open(stamp).read() == dig:
        return src, 0, "cached"
    cmd = [hipcc] + flags_for(src) + RESOURCE_REMARKS + ["-c", src, "-o", obj]
    r = subprocess.run(cmd, capture_output=True, text=True)
    lines = (r.stdout + r.stderr).splitlines()
    if r.returncode == 0:
        with open(obj + ".res.json", "w") as f:
            json.dump(parse_resource_remarks(lines), f, indent=0, sort_keys=True)
        with open(stamp, "w") as f:
            f.write(dig)
    # the HOST pass of a .hip file does not know the device feature named in FLAGS and says so once per pass: not a diagnostic;
    # the resource remarks went into the .res.json
    out = "\n".join(ln for ln in lines if "'-packed-fp32-ops' is not a recognized feature" not in ln and "-Rpass-analysis=kernel-resource-usage" not in ln
                    and not _REMARK_CONT.match(ln)).strip()
    return src, r.returncode, out


def build(force=False, verbose=True):
    hipcc = _hipcc()
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(os.path.dirname(HERE), "include", "tsii_hip.h"))
    jobs = []
    for src in sources():
        base = os.path.splitext(os.path.basename(src))[0]
        obj = os.path.join(OBJ, base + ".o")
        stamp = obj + ".sha"
        dig = _digest([src] + sorted(headers))
        if force and os.path.exists(stamp):
            os.remove(stamp)
        jobs.append((hipcc, src, obj, stamp, dig))
    with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
        results = list(ex.map(_compile_one, jobs))
    failed = [(s, out) for s, rc, out in results if rc != 0]
    for s, rc, out in results:
        if verbose and out and out != "cached":
            print(f"[build_ext] {os.path.basename(s)}:\n{out}", file=sys.stderr)
    if failed:
        raise RuntimeError("hipcc failed for: " + ", ".join(os.path.basename(s) for s, _ in failed))
    objs = [j[2] for j in jobs]
    relink = force or not os.path.exists(LIB) or any(os.path.getmtime(o) > os.path.getmtime(LIB) for o in objs)
    if relink:
        cmd = [hipcc, "-shared", "-fPIC", "--offload-arch=" + ARCH, "-o", LIB] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    if verbose:
        print(f"[build_ext] {LIB} ({'relinked' if relink else 'up to date'})")
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)

#!/usr/bin/env python3
"""Summarise rocprofv3 counter passes into the files kept under profiles/.

    python tools/pmc_summary.py hbm  <fetch_counter_collection.csv> <write_counter_collection.csv> <out_prefix> ["note"]
    python tools/pmc_summary.py sq   <sq_counter_collection.csv> <out.csv> ["note"]

`hbm`: per-kernel HBM traffic from two separate passes (--pmc FETCH_SIZE, --pmc WRITE_SIZE; counters in KiB).
On gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM section), so
reads are doubled: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024.  Writes <out_prefix>.csv and <out_prefix>.json
(the json is what bench.py reads for `roofline.traffic`).
`sq`: per-kernel sums of the SQ counters of one pass plus MFMA utilisation =
SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * SQ_BUSY_CYCLES / 32 SEs).
"""
import collections
import csv
import json
import os
import sys

ALL_FILES = os.environ.get("PMC_SUMMARY_ALL_FILES") == "1"   # fingerprint over every kernel file (a bf16-storage configuration)


def read_counters(path):
    """-> {kernel: {counter: [sum, launches]}}"""
    out = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
    with open(path, newline="") as f:
        for row in csv.DictReader(f):
            rec = out[row["Kernel_Name"]][row["Counter_Name"]]
            rec[0] += float(row["Counter_Value"])
            rec[1] += 1
    return out


def short(name):
    """'void tsii::k<...>(args)' -> 'tsii::k<...>'"""
    name = name.strip()
    if name.startswith("void "):
        name = name[5:]
    depth = 0
    for i, ch in enumerate(name):
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            return name[:i]
    return name


def hbm(fetch_csv, write_csv, prefix, note):
    fe, wr = read_counters(fetch_csv), read_counters(write_csv)
    rows = []
    for k in fe:
        f_sum, launches = fe[k].get("FETCH_SIZE", [0.0, 0])
        w_sum = wr.get(k, {}).get("WRITE_SIZE", [0.0, 0])[0]
        if launches == 0:
            continue
        rd = 2.0 * f_sum * 1024 / launches
        wb = w_sum * 1024 / launches
        rows.append((short(k), launches, f_sum, w_sum, rd, wb, rd + wb))
    rows.sort(key=lambda r: -r[6] * r[1])
    how = ("rocprofv3 --kernel-trace --pmc FETCH_SIZE and, in a separate pass, --pmc WRITE_SIZE. " + note +
           " Counters are in KiB; on gfx950 FETCH_SIZE reports half the bytes of wide coalesced reads "
           "(MI355X_MICROARCH.md, HBM section) so reads are doubled: bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024.")
    with open(prefix + ".csv", "w") as f:
        f.write("# " + how + "\n")
        f.write("Kernel,Launches,FETCH_SIZE_KiB_sum,WRITE_SIZE_KiB_sum,HBM_read_bytes_per_launch(2xFETCH),"
                "HBM_write_bytes_per_launch,HBM_bytes_per_launch\n")
        for r in rows:
            f.write('"%s",%d,%.0f,%.0f,%.0f,%.0f,%.0f\n' % r)
    import hashlib, os
    csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "text_segmentation_image_inpainting_amd", "csrc")
    h = hashlib.sha256()
    for fn in sorted(os.listdir(csrc)):
        if fn.endswith((".hip", ".h")) and (ALL_FILES or not fn.startswith("bf16_")):   # same rule as bench.csrc_sha
            h.update(open(os.path.join(csrc, fn), "rb").read())
    # fingerprint of the kernel sources this was measured at: bench.py reports `roofline.traffic` only while it matches
    js = {"_how": how, "csrc_sha": h.hexdigest()[:16], "kernels": {r[0]: {"launches": r[1], "read_bytes_per_launch": r[4], "write_bytes_per_launch": r[5],
                                          "bytes_per_launch": r[6]} for r in rows}}
    with open(prefix + ".json", "w") as f:
        json.dump(js, f, indent=1)


def sq(path, out, note):
    data = read_counters(path)
    names = ["SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY",
             "SQ_ACTIVE_INST_ANY", "SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE"]
    rows = []
    for k, c in data.items():
        vals = [c.get(n, [0.0, 0])[0] for n in names]
        busy = vals[1]
        util = vals[2] / (1024.0 * busy / 32.0) if busy > 0 else 0.0
        rows.append((short(k), vals, util))
    rows.sort(key=lambda r: -r[1][1])
    with open(out, "w") as f:
        f.write("# rocprofv3 --kernel-trace --pmc " + " ".join(names) + " ; " + note + " ; sums over all launches\n")
        f.write("# MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs * SQ_BUSY_CYCLES/32 SEs)\n")
        f.write("Kernel," + ",".join(names) + ",MfmaUtil\n")
        for k, vals, util in rows:
            f.write('"%s",' % k + ",".join("%.0f" % v for v in vals) + ",%.4f\n" % util)


if __name__ == "__main__":
    mode = sys.argv[1]
    if mode == "hbm":
        hbm(sys.argv[2], sys.argv[3], sys.argv[4], sys.argv[5] if len(sys.argv) > 5 else "")
    elif mode == "sq":
        sq(sys.argv[2], sys.argv[3], sys.argv[4] if len(sys.argv) > 4 else "")
    else:
        raise SystemExit(__doc__)

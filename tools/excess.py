#!/usr/bin/env python3
"""Rank the launches of a per-shape table (tools/profile_step.py output) by EXCESS time: measured - max(algorithmic bytes / 5.5 TB/s,
6-product matrix time at 60 % of the bf16 peak) with bench.py's traffic model.    python tools/excess.py <per_shape.log> [rows]"""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

def main():
    rows = []
    for ln in open(sys.argv[1]):
        m = re.match(r"(tsii_\w+)\s+x(\d+)\s+([\d.]+) ms.*args=\(([^)]*)\)", ln)
        if not m or m.group(1) not in bench.CALLS:
            continue
        a = tuple(float(v) if ("." in v or "e" in v) else int(v) for v in (s.strip() for s in m.group(4).split(",")) if v)
        cls, fn = bench.CALLS[m.group(1)]
        try:
            by, macs = fn(a)
        except Exception:
            continue
        cnt, ms = int(m.group(2)), float(m.group(3))
        floor = max(by / 5.5e12, (2 * macs * 6 / (0.6 * 2.5e15)) if cls.startswith(("gemm", "dense")) else 0.0) * 1e3 * cnt
        rows.append((ms - floor, ms, floor, cnt, m.group(1), a[:14]))
    rows.sort(reverse=True)
    tot = sum(r[1] for r in rows)
    print(f"modelled launches: {tot:.1f} ms, practical floors {sum(r[2] for r in rows):.1f} ms")
    for ex, ms, fl, cnt, name, a in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
        print(f"excess {ex:7.3f} ms  ({ms:7.3f} measured, {fl:6.3f} practical floor) x{cnt:<2d} {name[5:]:18s} {a}")

if __name__ == "__main__":
    main()

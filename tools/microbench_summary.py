#!/usr/bin/env python3
"""Turns the output of tools/microbench.hip (run on the GPU box) into the committed summary bench.py reads its integer
peak from:   python tools/microbench_summary.py <microbench stdout> profiles/r03_microbench_int_rates.json
The summary carries the sha256 of the microbenchmark source: bench.py ignores it once the source changes."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    text = open(sys.argv[1]).read()
    m = re.search(r"device (\S+), (\d+) CUs, clock ([0-9.]+) GHz", text)
    dev, cus, clk = m.group(1), int(m.group(2)), float(m.group(3))
    row = re.search(r"v_mad_u64_u32 distinct srcs, 8 chains.*?4 w/SIMD:\s*([0-9.]+) cyc/inst", text)
    cyc = float(row.group(1))
    src = open(os.path.join(ROOT, "tools", "microbench.hip"), "rb").read()
    out = {"device": dev, "cus": cus, "clock_ghz": clk, "cycles_per_v_mad_u64_u32": cyc,
           "peak_Tmad_per_s": round(cus * 4 * 64 * clk / cyc / 1e3, 2),
           "row": "v_mad_u64_u32 distinct srcs, 8 chains, 4 waves/SIMD", "src_sha256": hashlib.sha256(src).hexdigest(),
           "raw": text.strip().splitlines()}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    print({k: out[k] for k in ("cus", "clock_ghz", "cycles_per_v_mad_u64_u32", "peak_Tmad_per_s")})


if __name__ == "__main__":
    main()

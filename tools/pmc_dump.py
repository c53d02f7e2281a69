#!/usr/bin/env python3
"""Per-dispatch counter table of the LAST proof of a `rocprofv3 --kernel-trace --pmc ...` run (bench.py, serial-MSM mode):
    python tools/pmc_dump.py <rocprofv3 output dir> [name filter]"""
import collections
import csv
import glob
import os
import re
import sys


def main():
    f = glob.glob(os.path.join(sys.argv[1], "**", "*counter_collection.csv"), recursive=True)[0]
    flt = sys.argv[2] if len(sys.argv) > 2 else ""
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        e = per.setdefault(k, {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]),
                               "dur_us": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    rows = list(per.values())
    idx = [i for i, r in enumerate(rows) if "qap_eval" in r["name"]]
    rows = rows[idx[-1]:] if idx else rows
    for r in rows:
        n = re.sub(r"\(.*", "", re.sub(r"^void ", "", r["name"]).replace("g16::", ""))
        if flt and flt not in n:
            continue
        extra = " ".join(f"{k}={v:.4g}" for k, v in r.items() if k not in ("name", "grid", "dur_us"))
        print(f"{n:52s} grid={r['grid']:8d} dur_us={r['dur_us']:8.1f} {extra}")


if __name__ == "__main__":
    main()

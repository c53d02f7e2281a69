#!/usr/bin/env python3
"""gpurun_out/final/ (tools/profile_round.sh) -> profiles/<tag>_pmc_traffic.json, <tag>_pmc_accumulate.txt,
<tag>_final_kernel_stats.csv, <tag>_final_bench_n1.json.   usage: python tools/pmc_summary.py [tag] [dir]"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "gpurun_out", "final")
prof = os.path.join(ROOT, "profiles")


def counters(d):
    f = glob.glob(os.path.join(src, d, "**", "*counter_collection.csv"), recursive=True)[0]
    per = collections.OrderedDict()
    for r in csv.DictReader(open(f)):
        k = int(r["Dispatch_Id"])
        e = per.setdefault(k, {"name": r["Kernel_Name"], "grid": int(r["Grid_Size"]),
                               "dur_ms": (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    return list(per.values())


def short(n):
    n = re.sub(r"^void ", "", n)
    n = n.replace("g16::", "")
    return re.sub(r"\(.*", "", n)


def last_proof(rows):
    idx = [i for i, r in enumerate(rows) if "qap_eval" in r["name"]]
    return rows[idx[-1]:]


shutil.copy(os.path.join(src, "bench.json"), os.path.join(prof, f"{tag}_final_bench_n1.json"))
stats = glob.glob(os.path.join(src, "prof", "**", "*kernel_stats.csv"), recursive=True)
if stats:
    shutil.copy(stats[0], os.path.join(prof, f"{tag}_final_kernel_stats.csv"))

p1, p2, p3 = last_proof(counters("pmc1")), last_proof(counters("pmc2")), last_proof(counters("pmc3"))
acc1 = [r for r in p1 if "msm_accumulate_kernel<g16::Fq29Ops>" in r["name"]]
acc2 = [r for r in p2 if "msm_accumulate_kernel<g16::Fq29Ops>" in r["name"]]
fetch = [int(r["FETCH_SIZE"] * 1024) for r in acc1]
write = [int(r["WRITE_SIZE"] * 1024) for r in acc2]
miss = [int(r["TCC_MISS_sum"] * 64) for r in acc2]
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_costs  # noqa: E402
bench = json.load(open(os.path.join(src, "bench.json")))
cfgw = bench["config"]["workload"]
m = re.search(r"nVars=(\d+), nConstraints=(-?\d+)", cfgw)
circuit = "nzcp_live" if cfgw.startswith("nzcp_live") else "nzcp_example" if cfgw.startswith("nzcp_example") else "synthetic"
traffic = {
    "kernel": "msm_accumulate_kernel<Fq29Ops>",
    "kernel_src_sha256": kernel_costs.kernel_source_hash(),
    "workload_id": f"{circuit}:{m.group(1)}:{m.group(2)}",
    "launches": "the two G1 launches of one proof in launch order (fused witness group A+B1+C, then H), serial-MSM mode "
                "(G16_SERIAL_MSM=1), default bench workload",
    "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "tcc_miss_x64B_per_launch": miss,
    "avg_traffic_bytes_per_launch": int(sum(f + w for f, w in zip(fetch, write)) / max(1, len(fetch))),
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in separate passes with "
              "--kernel-trace (MI355X_MICROARCH.md HBM section); counters are KiB -> x1024. FETCH_SIZE is reported RAW: "
              "the guide's x2 correction is calibrated for 16 B/lane coalesced streams only; these are 64-byte random "
              "gathers (4 x dwordx4 per lane, one aligned 64-byte sector per point), calibrated on a known byte count in "
              "this very pattern (tools/fetch_calib.hip, profiles/r02_fetch_size_calibration.txt: reading / true bytes = "
              "1.000 for random 64-byte gathers, 0.500 for the same loads on consecutive records and for 16 B/lane "
              "streams); TCC_MISS_sum x 64 B gives the same figure within ~5 %.",
    "source": f"profiles/{tag}_pmc_accumulate.txt (tools/profile_round.sh + tools/pmc_summary.py)",
}
json.dump(traffic, open(os.path.join(prof, f"{tag}_pmc_traffic.json"), "w"), indent=1)

with open(os.path.join(prof, f"{tag}_pmc_accumulate.txt"), "w") as o:
    o.write("# rocprofv3 --pmc, serial-MSM mode (G16_SERIAL_MSM=1), python3 bench.py --steps 2 --warmup 1 --no-cpu "
            "--batch-streams 0; rows = dispatches of the last proof (tools/profile_round.sh)\n")
    o.write("# pass 3: SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU "
            "SQ_INSTS_VALU SQ_WAVES\n")
    # SQ_WAVE_CYCLES per wave, in units calibrated on the persistent H accumulate grid (its waves live as long as the
    # kernel): wave_life = average wave lifetime / kernel duration.  A tail kernel with wave_life well below 1 lasts as long
    # as its LONGEST dependency chain, so valu_insts_per_wave (an average) and simd_cycles_per_valu_inst understate what its
    # slowest waves do (r03: profiles/r03_sweeps.txt section 2).
    hacc = [r for r in p3 if "msm_accumulate_kernel<g16::Fq29Ops>" in r["name"]]
    unit = max((r.get("SQ_WAVE_CYCLES", 0) / max(r.get("SQ_WAVES", 1), 1) / r["dur_ms"] for r in hacc), default=0.0)
    o.write("# wave_life = (SQ_WAVE_CYCLES / SQ_WAVES) / kernel duration, normalised to the persistent H accumulate grid (= 1.00)\n")
    for r in p3:
        if not r["name"].startswith(("void g16::msm_", "g16::ntt", "g16::qap", "g16::msm_", "void g16::")):
            continue
        w, wc = max(r.get("SQ_WAVES", 1), 1), max(r.get("SQ_WAVE_CYCLES", 1), 1)
        iv = r.get("SQ_INSTS_VALU", 0)
        life = (wc / w / r["dur_ms"] / unit) if unit and r["dur_ms"] else 0.0
        o.write(f"{short(r['name']):44s} grid={r['grid']:8d} dur_ms={r['dur_ms']:.3f} waves={int(w)} "
                f"valu_insts_per_wave={iv / w:.0f} simd_cycles_per_valu_inst={r['dur_ms'] * 1e-3 * 2.4e9 / max(iv / 1024, 1):.2f} "
                f"wait_mem/wave_cyc={r.get('SQ_WAIT_ANY', 0) / wc:.2f} issue_stall/wave_cyc={r.get('SQ_WAIT_INST_ANY', 0) / wc:.2f} "
                f"active/wave_cyc={r.get('SQ_ACTIVE_INST_ANY', 0) / wc:.2f} wave_life={life:.2f}\n")
    o.write("# pass 1: FETCH_SIZE (KiB, raw)   pass 2: WRITE_SIZE (KiB), TCC_HIT_sum, TCC_MISS_sum\n")
    for a, b in zip(p1, p2):
        if any(k in a["name"] for k in ("accumulate", "ntt_pass", "ntt_mid", "ntt_last", "bin_pass", "bin_direct", "bin_sort", "qap_eval", "qap_long")):
            o.write(f"{short(a['name']):44s} grid={a['grid']:8d} dur_ms={a['dur_ms']:.3f} FETCH_SIZE={a.get('FETCH_SIZE', 0):.4g} "
                    f"WRITE_SIZE={b.get('WRITE_SIZE', 0):.4g} TCC_HIT={b.get('TCC_HIT_sum', 0):.4g} TCC_MISS={b.get('TCC_MISS_sum', 0):.4g}\n")
print(json.dumps(traffic, indent=1))

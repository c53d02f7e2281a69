#!/usr/bin/env python3
"""Concurrency summary of g16_prove_batch from a rocprofv3 --kernel-trace CSV (tools/batch_timeline.sh):
steady-state period per proof, share of the time with any kernel / with a chip-filling kernel running, and how far
consecutive proofs overlap (a proof = qap_eval_kernel start .. the last msm_row_final / to_canon kernel before the
next-but-N qap_eval on the same hardware queue set is not recoverable from the trace, so the overlap is measured as
the number of accumulate kernels of OTHER proofs that start inside a proof's QAP..H-accumulate span)."""
import csv, re, sys, collections

def main(path, last=20):
    rows = list(csv.DictReader(open(path)))
    for r in rows:
        r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].replace("void ", "").replace("g16::", "").replace("(anonymous namespace)::", "")
        r["n"] = re.sub(r"\(.*", "", n)
    rows.sort(key=lambda r: r["s"])
    q = [r for r in rows if r["n"].startswith("qap_eval_kernel")]
    if len(q) < last + 2:
        print("too few proofs in the trace"); return
    w0, w1, nproof = q[-last]["s"], q[-2]["s"], last - 2
    sel = [r for r in rows if r["e"] > w0 and r["s"] < w1]
    def union(iv):
        t, cs, ce = 0, None, None
        for s, e in sorted(iv):
            s, e = max(s, w0), min(e, w1)
            if e <= s: continue
            if cs is None: cs, ce = s, e
            elif s <= ce: ce = max(ce, e)
            else: t += ce - cs; cs, ce = s, e
        return t + (ce - cs if cs is not None else 0)
    W = w1 - w0
    tp = ("msm_accumulate", "ntt_", "qap_", "msm_bin_pass")
    print(f"steady state over {nproof} proofs: {W / 1e6 / nproof:.3f} ms per proof (under the profiler)")
    print(f"  some kernel running            {union([(r['s'], r['e']) for r in sel]) / W:.3f} of the time")
    print(f"  a chip-filling kernel running  {union([(r['s'], r['e']) for r in sel if r['n'].startswith(tp)]) / W:.3f}   (accumulate, NTT, QAP, bin passes)")
    # overlap: H accumulate kernels (the longest G1 accumulate of a proof) that run while ANOTHER proof's NTT runs
    ntt = [(r["s"], r["e"]) for r in sel if r["n"].startswith("ntt_")]
    acc = [(r["s"], r["e"]) for r in sel if r["n"].startswith("msm_accumulate_kernel<Fq29Ops>") and r["e"] - r["s"] > 600e3]
    both = 0
    for s, e in acc:
        both += sum(max(0, min(e, e2) - max(s, s2)) for s2, e2 in ntt)
    print(f"  H accumulate overlapped by another proof's NTT kernels: {both / max(1, sum(e - s for s, e in acc)):.2f} of its duration")
    agg = collections.defaultdict(lambda: [0, 0])
    for r in sel:
        a = agg[r["n"]]; a[0] += min(r["e"], w1) - max(r["s"], w0); a[1] += 1
    for n, (d, c) in sorted(agg.items(), key=lambda x: -x[1][0])[:12]:
        print(f"    {n[:52]:52s} {d / 1e6 / nproof:6.3f} ms/proof  {c / nproof:5.1f} calls/proof")

if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 20)

#!/usr/bin/env python3
"""Static instruction counts of the dominant kernel, generated at build time from the compiler's own gfx950
assembly (so that bench.py's integer roofline never carries a stale hand-written constant):
    python tools/kernel_costs.py <msm_g1.s> <out.json>
Counts v_mad_u64_u32 / VALU instructions between the label of msm_accumulate_kernel<Fq29Ops> and its s_endpgm: the
kernel holds exactly one inlined mixed addition (x29_madd_fast), so the static counts are per loop iteration."""
import hashlib
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "nzcp-circom_amd", "csrc")


def kernel_source_hash():
    """sha256 over the kernel sources: what a PMC profile must have been taken on to describe the current build."""
    h = hashlib.sha256()
    # the translation unit of the roofline kernel (msm_accumulate_kernel<G1>) and every header it includes -- not the
    # whole library: an edit to the verifier or the PLONK prover does not change what the PMC passes measured
    for name in ("bn254_consts.h", "ec.cuh", "ec29.cuh", "fp.cuh", "fq29.cuh", "internal.h", "msm.cuh", "msm_g1.hip"):
        h.update(name.encode())
        h.update(open(os.path.join(CSRC, name), "rb").read())
    return h.hexdigest()


def main():
    asm, out = sys.argv[1], sys.argv[2]
    text = open(asm).read()
    m = re.search(r"^(_ZN3g1621msm_accumulate_kernelINS_7Fq29OpsEE\w*):.*$", text, re.M)
    if not m:
        raise SystemExit("kernel label not found")
    body = text[m.end():]
    body = body[:body.index("s_endpgm")]
    lines = [ln.strip() for ln in body.splitlines()]
    insts = [ln for ln in lines if ln and not ln.startswith((";", ".", "//")) and not ln.endswith(":")]
    mads = sum(1 for ln in insts if ln.startswith("v_mad_u64_u32"))
    valu = sum(1 for ln in insts if ln.startswith("v_"))
    res = {"kernel": "msm_accumulate_kernel<Fq29Ops>", "static_v_mad_u64_u32": mads, "static_valu": valu,
           "static_instructions": len(insts), "kernel_src_sha256": kernel_source_hash(),
           "note": "one inlined x29_madd_fast per loop iteration: counts are per bucket addition (plus loop bookkeeping)"}
    json.dump(res, open(out, "w"), indent=1)
    print(res)


if __name__ == "__main__":
    main()

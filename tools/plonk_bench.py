#!/usr/bin/env python3
"""PLONK prover bench: proofs/s of `snarkjs plonk prove`'s device replacement (csrc/plonk.hip) on the
REAL NZCP circuit -- NZCPPubIdentity built natively (g16_nzcp_circuit_setup), converted to PLONK gates and keyed with
a known tau by g16_plonk_setup (the stand-in for `snarkjs plonk setup ... powersOfTau28_hez_final_22.ptau`,
/root/reference/Makefile:31).  Every timed proof uses fresh random
blinding; one proof is checked by the oracle's KZG verifier (oracle/plonk.py) against the key's verification key.

    python tools/plonk_bench.py [--circuit nzcp_example|nzcp_live] [--steps K] [--warmup W]
Prints one JSON line."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--circuit", choices=["nzcp_example", "nzcp_live"], default="nzcp_example")
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--no-verify", action="store_true")
    a = ap.parse_args()
    amd = entry.load_package()
    amd.load()
    entry.oracle_path()
    import formats as f
    import plonk as pk
    import synth
    seed = synth.SEED_NZCP

    def log(m):
        print(f"[plonk bench] {m}", file=sys.stderr, flush=True)
    t0 = time.time()
    import nzcp_pass
    live = a.circuit == "nzcp_live"
    params = amd.NZCP_LIVE_PARAMS if live else amd.NZCP_EXAMPLE_PARAMS
    tbs = nzcp_pass.to_be_signed("Jack", "Sparrow", "1960-04-16", live=live, exp=1951416330)
    out = amd.nzcp_circuit_setup(params, tbs, seed, 0, want_zkey=False, want_r1cs=True)
    r1cs, wtns = out["r1cs"], out["wtns"]
    log(f"{a.circuit}: {out['n_constraints']} R1CS constraints, r1cs {len(r1cs) / 1e6:.0f} MB, built in {time.time() - t0:.1f}s")
    t0 = time.time()
    zkey = amd.plonk_setup(r1cs, seed, device=0, with_lagrange=False)
    del r1cs
    log(f"plonk setup (known tau, no Lagrange section): zkey {len(zkey) / 1e9:.2f} GB in {time.time() - t0:.1f}s")
    vk = None if a.no_verify else pk.vkey_from_zkey(zkey)
    t0 = time.time()
    prover = amd.PlonkProver(zkey, device=0)
    del zkey
    log(f"g16_plonk_create {time.time() - t0:.1f}s: domain 2^{prover.domain_size.bit_length() - 1}, {prover.n_constraints} gates, "
        f"{prover.n_additions} additions in {prover.levels} dependency levels, {prover.n_public} public signals")
    proof = pub = None
    for _ in range(a.warmup):
        proof, pub = prover.prove(wtns)
    ts = []
    for _ in range(a.steps):
        t = time.perf_counter()
        pr, pb = prover.prove_raw(wtns)
        ts.append(time.perf_counter() - t)
    proof, pub = prover.prove(wtns)
    verified = None
    if vk is not None:
        t0 = time.time()
        verified = pk.verify(vk, [int(x) for x in pub], pk.proof_from_obj(proof))
        log(f"oracle KZG verifier on the full-size proof: {verified} ({time.time() - t0:.1f}s)")
        assert verified
    ts.sort()
    med = ts[len(ts) // 2]
    print(json.dumps({"metric": "plonk_proofs_per_sec", "value": round(1 / med, 3), "unit": "proofs/s", "n_gpus": 1,
                      "ms_per_proof_p50": round(med * 1e3, 2), "ms_per_proof_min": round(ts[0] * 1e3, 2), "steps": a.steps,
                      "config": {"workload": f"{a.circuit}: NZCPPubIdentity{tuple(params)} as PLONK, domain 2^{prover.domain_size.bit_length() - 1}, "
                                             f"{prover.n_constraints} gates ({prover.n_additions} addition gates), {prover.n_public} public signals"},
                      "data": "real NZCP constraint system built natively, test-only setup with a known tau",
                      "rounds_ms": prover.timings(), "verified_by_oracle": verified}))
    prover.close()


if __name__ == "__main__":
    main()

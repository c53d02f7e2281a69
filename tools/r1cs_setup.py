#!/usr/bin/env python3
"""Test-only trapdoor Groth16 setup of a real circuit:  circom's .r1cs  ->  snarkjs-layout .zkey +
verification_key.json, through the product's g16_r1cs_setup (host code; no GPU needed).

    python tools/r1cs_setup.py nzcp_liveTest.r1cs nzcp_live_test.zkey verification_key.json [--seed N]

Not a ceremony: the trapdoor is derived from --seed and therefore known.  It exists so that the
REAL nzcp_live constraint system (circom --r1cs, /root/reference/Makefile:8-9) can be proved and
benchmarked with this prover on a box that has circom but no ptau/phase-2 files:
    node nzcp-circom_amd/js/cli.js groth16 prove nzcp_live_test.zkey witness.wtns proof.json public.json
    snarkjs groth16 verify verification_key.json public.json proof.json
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as entry  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("r1cs")
    ap.add_argument("zkey")
    ap.add_argument("vkey_json")
    ap.add_argument("--seed", type=int, default=0x6E7A6370)
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--device", type=int, default=-1,
                    help="run the fixed-base multiplications on this GPU (g16_setup_device); default: host threads")
    a = ap.parse_args()
    amd = entry.load_package()
    amd.setup_device(a.device)
    data = open(a.r1cs, "rb").read()
    zkey, vkey = amd.r1cs_setup(data, a.seed, a.threads)
    n_public = None
    # nPublic: header section 2 of the zkey = n8q q n8r r nVars nPublic domainSize
    pos = 12
    while True:
        sid = int.from_bytes(zkey[pos:pos + 4], "little")
        size = int.from_bytes(zkey[pos + 4:pos + 12], "little")
        if sid == 2:
            n_public = int.from_bytes(zkey[pos + 12 + 76:pos + 12 + 80], "little")
            break
        pos += 12 + size
    open(a.zkey, "wb").write(zkey)
    open(a.vkey_json, "w").write(json.dumps(amd.vkey_json(vkey, n_public), indent=1))
    print(f"{a.zkey}: {len(zkey)} bytes, nPublic = {n_public}; {a.vkey_json} written (trapdoor seed {a.seed})")


if __name__ == "__main__":
    main()

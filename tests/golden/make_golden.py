#!/usr/bin/env python3
"""Regenerates the golden fixtures in this directory with the PYTHON ORACLE ONLY
(oracle/synth.py -> oracle/groth16.py setup/prove -> oracle/formats.py writers).

The reference repo has no prove-boundary fixtures (/root/reference/.gitignore:2-4
excludes zkey/wtns; no proof.json anywhere), so these vectors are produced by the restatement and
pinned three ways before being written: proof == trapdoor known-answer ([a]G1,[b]G2,[c]G1),
pairing check passes, and files round-trip through the readers.

Per case <name>: <name>.zkey, <name>.wtns, <name>.json = {r, s, proof, public, vkey, kat:{a,b,c}}.
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import formats as f  # noqa: E402
import groth16 as g  # noqa: E402
import synth  # noqa: E402

CASES = [  # name, nVars, nPublic, nConstraints, seed
    ("tiny", 24, 2, 12, 1),
    ("small", 150, 6, 120, 2),
    ("nzcp513", 700, 513, 480, synth.SEED_NZCP),   # p = 513 like NZCPPubIdentity, N = 1024
]


def main():
    for name, n, p, m, seed in CASES:
        rows, w = synth.make(n, p, m, seed)
        assert synth.check_r1cs(rows, w)
        zk, sec = g.setup(n, p, rows, g.trapdoor(seed + 1))
        rng = synth.Xoshiro(seed + 2)
        r, s = rng.rand_fr(), rng.rand_fr()
        proof, pub = g.prove(zk, w, r, s)
        assert proof == g.expected_proof(sec, p, w, r, s), "trapdoor KAT failed"
        assert g.verify(zk, pub, proof), "pairing check failed"
        zb, wb = f.write_zkey(zk), f.write_wtns(w)
        assert f.read_zkey(zb)["H"] == zk["H"] and f.read_wtns(wb)["w"] == w
        a, b, c = g.expected_proof_scalars(sec, p, w, r, s)
        meta = {"n": n, "p": p, "m": m, "seed": seed, "r": str(r), "s": str(s),
                "proof": f.proof_obj(*proof), "public": [str(x) for x in pub],
                "vkey": f.vkey_obj(zk), "kat": {"a": str(a), "b": str(b), "c": str(c)}}
        open(os.path.join(HERE, name + ".zkey"), "wb").write(zb)
        open(os.path.join(HERE, name + ".wtns"), "wb").write(wb)
        open(os.path.join(HERE, name + ".json"), "w").write(json.dumps(meta, indent=1))
        print(name, "zkey", len(zb), "wtns", len(wb), "domain", zk["domainSize"])


if __name__ == "__main__":
    main()

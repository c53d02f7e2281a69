#!/usr/bin/env python3
"""Regenerates the PLONK golden fixture of this directory with the PYTHON ORACLE ONLY (oracle/plonk.py).

The reference holds no PLONK artefact either (its Makefile:30-33 scripts `snarkjs plonk setup` and the key exports; no
zkey, proof or verification key is committed), so the vectors are the restatement's, pinned before being written by its
independently derived KZG verifier and by a negative case.

plonk_small.zkey (snarkjs PLONK layout, with the Lagrange section), plonk_small.wtns, plonk_small.json =
{n, p, m, seed, tau, blinding: [b1..b9], proof, public, vkey}."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
import bn254 as b  # noqa: E402
import formats as f  # noqa: E402
import plonk as pk  # noqa: E402
import synth  # noqa: E402


def vkey_json(zk):
    def g1(P):
        return ["0", "1", "0"] if P is None else [str(P[0]), str(P[1]), "1"]
    X2 = zk["X_2"]
    vk = {"protocol": "plonk", "curve": "bn128", "nPublic": zk["nPublic"], "power": zk["power"], "k1": str(zk["k1"]), "k2": str(zk["k2"])}
    for k in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
        vk[k] = g1(zk[k])
    vk["X_2"] = [[str(X2[0][0]), str(X2[0][1])], [str(X2[1][0]), str(X2[1][1])], ["1", "0"]]
    vk["w"] = str(b.fr_root(zk["power"]))
    return vk


def main():
    n, p, m, seed, tau = 40, 3, 24, 11, 0x706c6f6e6b
    rows, w = synth.make(n, p, m, seed)
    zk = pk.setup(n, p, rows, tau)
    rng = synth.Xoshiro(seed + 5)
    bl = {i: rng.rand_fr() for i in range(1, 10)}
    proof, pub = pk.prove(zk, w, bl)
    vk = pk.vkey(zk)
    assert pk.verify(vk, pub, proof), "oracle verifier rejected its own proof"
    bad = list(pub)
    bad[0] = (bad[0] + 1) % b.R
    assert not pk.verify(vk, bad, proof)
    zb, wb = pk.write_zkey(zk), f.write_wtns(w)
    assert pk.vkey_from_zkey(zb)["Qm"] == zk["Qm"]
    meta = {"n": n, "p": p, "m": m, "seed": seed, "tau": str(tau), "blinding": [str(bl[i]) for i in range(1, 10)],
            "proof": pk.proof_obj(proof), "public": [str(x) for x in pub], "vkey": vkey_json(zk)}
    open(os.path.join(HERE, "plonk_small.zkey"), "wb").write(zb)
    open(os.path.join(HERE, "plonk_small.wtns"), "wb").write(wb)
    open(os.path.join(HERE, "plonk_small.json"), "w").write(json.dumps(meta, indent=1))
    print("plonk_small zkey", len(zb), "wtns", len(wb), "domain", zk["domainSize"])


if __name__ == "__main__":
    main()

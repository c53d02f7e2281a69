#!/usr/bin/env python3
"""Workload for `rocprofv3 --kernel-trace --stats -- python3 tools/verify_profile.py`: the Groth16 batch verifier on
1 024 proofs (513 public signals each) and the PLONK batch verifier on 256 proofs, small keys from the oracle-free
product setup paths.  Prints the host-side wall times."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

amd = entry.load_package()
amd.load()
entry.oracle_path()
import formats as f  # noqa: E402
import plonk as pk  # noqa: E402
import synth  # noqa: E402

n, p, m, seed = 900, 513, 380, 24
zkey, wtns, vkey = amd.synth_setup(n, p, m, seed, 4)
prover = amd.Prover(zkey)
proof, pub = prover.prove(wtns)
prover.close()
ver = amd.Verifier(vkey, p, montgomery=True)
items = [(pub, proof)] * 1024
ver.verify_batch(items[:64])
t = time.perf_counter()
ok = ver.verify_batch(items)
print(f"groth16 verify: 1024 proofs in {1e3 * (time.perf_counter() - t):.1f} ms (incl. Python marshalling), all ok {all(ok)}; device phases {ver.timings()}")
ver.close()
rows, w = synth.make(n, p, m, seed)
_, rows_c, _ = synth.gen_circuit(n, p, m, seed)
pz = amd.plonk_setup(f.write_r1cs(n, p, 0, rows_c), seed, device=0)
vk = pk.vkey_from_zkey(pz)
pp = amd.PlonkProver(pz)
pproof, ppub = pp.prove(wtns)
pp.close()


def g1j(P):
    return ["0", "1", "0"] if P is None else [str(P[0]), str(P[1]), "1"]


vkj = {"protocol": "plonk", "nPublic": vk["nPublic"], "power": vk["power"], "k1": vk["k1"], "k2": vk["k2"],
       "X_2": [[str(vk["X_2"][0][0]), str(vk["X_2"][0][1])], [str(vk["X_2"][1][0]), str(vk["X_2"][1][1])], ["1", "0"]]}
for k in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
    vkj[k] = g1j(vk[k])
pv = amd.PlonkVerifier(vkj)
pitems = [(ppub, pproof)] * 256
pv.verify_batch(pitems[:16])
t = time.perf_counter()
pok = pv.verify_batch(pitems)
print(f"plonk verify: 256 proofs in {1e3 * (time.perf_counter() - t):.1f} ms (incl. Python marshalling), all ok {all(pok)}")
pv.close()

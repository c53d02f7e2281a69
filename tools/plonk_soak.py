import sys, os, time, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import __graft_entry__ as e
amd = e.load_package(); amd.load(); e.oracle_path()
sys.path.insert(0, os.path.join(e.ROOT, "tools"))
import plonk as pk, synth, formats as f, nzcp_pass
# PLONK soak: one key, 40 proofs with fresh blinding over 4 different passes, every proof through the device verifier
params = amd.NZCP_EXAMPLE_PARAMS
names = [("Jack", "Sparrow", "1960-04-16"), ("Li", "Wei", "2001-02-03"), ("Aroha", "Ngata", "1975-07-21"), ("Olivia", "Smith", "1999-05-27")]
outs = [amd.nzcp_circuit_setup(params, nzcp_pass.to_be_signed(g, fa, d, live=False, exp=1951416330 - i), 7, 0, want_zkey=False, want_r1cs=(i == 0))
        for i, (g, fa, d) in enumerate(names)]
zkey = amd.plonk_setup(outs[0]["r1cs"], 7, device=0, with_lagrange=False)
vk = pk.vkey_from_zkey(zkey)
prover = amd.PlonkProver(zkey); del zkey
def g1j(P): return ["0", "1", "0"] if P is None else [str(P[0]), str(P[1]), "1"]
vkj = {"protocol": "plonk", "nPublic": vk["nPublic"], "power": vk["power"], "k1": vk["k1"], "k2": vk["k2"],
       "X_2": [[str(vk["X_2"][0][0]), str(vk["X_2"][0][1])], [str(vk["X_2"][1][0]), str(vk["X_2"][1][1])], ["1", "0"]]}
for k in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"): vkj[k] = g1j(vk[k])
ver = amd.PlonkVerifier(vkj)
items = []
t0 = time.time()
for i in range(40):
    proof, pub = prover.prove(outs[i % 4]["wtns"])
    items.append((pub, proof))
t1 = time.time()
ok = ver.verify_batch(items)
print(json.dumps({"plonk_soak": {"proofs": 40, "distinct_passes": 4, "all_verified_on_device": all(ok), "seconds": round(t1 - t0, 2),
                                 "oracle_spot_check": pk.verify(vk, [int(x) for x in items[7][0]], pk.proof_from_obj(items[7][1]))}}))
assert all(ok)

"""oracle/plonk.py on its own (CPU): Keccak-256 known answers, the R1CS -> PLONK gate conversion (every gate holds on
the extended witness), and prove -> verify with the independently derived KZG verifier, including rejections."""
import pytest

import bn254 as b
import plonk as pk
import synth


def test_keccak256_known_answers():
    assert pk.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert pk.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    # one byte short of / exactly / one byte past the 136-byte rate
    assert pk.keccak256(b"\x00" * 135).hex() != pk.keccak256(b"\x00" * 136).hex() != pk.keccak256(b"\x00" * 137).hex()
    assert pk.keccak256(b"The quick brown fox jumps over the lazy dog").hex() == \
        "4d741b6f1eb29cb2a9b9911c82f56fa8d73b04959d3d9d222895df6c0b28aa15"


@pytest.mark.parametrize("n,p,m,seed", [(24, 2, 12, 1), (60, 5, 40, 3)])
def test_gates_hold_and_proof_verifies(n, p, m, seed):
    rows, w = synth.make(n, p, m, seed)
    gates, adds, pnv = pk.r1cs_to_plonk(n, p, rows)
    assert pnv == n + len(adds)
    we = pk.extend_witness(w, adds)
    assert pk.check_gates(gates, p, we)
    bad = list(we)
    bad[n - 1] = (bad[n - 1] + 1) % b.R
    assert not pk.check_gates(gates, p, bad)
    zk = pk.setup(n, p, rows, tau=0x1234567 + seed)
    rng = synth.Xoshiro(seed + 40)
    bl = {i: rng.rand_fr() for i in range(1, 10)}
    proof, pub = pk.prove(zk, w, bl)
    vk = pk.vkey(zk)
    assert pub == [x % b.R for x in w[1:p + 1]]
    assert pk.verify(vk, pub, proof)
    pub2 = list(pub)
    pub2[-1] = (pub2[-1] + 1) % b.R
    assert not pk.verify(vk, pub2, proof)
    for k in ("eval_a", "eval_zw", "eval_r"):
        t = dict(proof)
        t[k] = (t[k] + 1) % b.R
        assert not pk.verify(vk, pub, t)
    t = dict(proof)
    t["Z"] = proof["A"]
    assert not pk.verify(vk, pub, t)
    # other blinding: another proof of the same statement, verifies as well
    proof2, _ = pk.prove(zk, w, {i: rng.rand_fr() for i in range(1, 10)})
    assert proof2["A"] != proof["A"] and pk.verify(vk, pub, proof2)
    # the JSON shape round-trips
    assert pk.proof_from_obj(pk.proof_obj(proof)) == proof
    z = pk.write_zkey(zk)
    assert z[:4] == b"zkey" and len(pk.write_zkey(zk, with_lagrange=False)) < len(z)


def test_oracle_reproduces_the_plonk_golden_fixture():
    """tests/golden/plonk_small.* (make_golden_plonk.py): the oracle still produces the committed key, proof and
    verification key -- a change of the restatement shows up here, not only in the GPU parity tests."""
    import json
    import formats as f
    from conftest import golden_path
    meta = json.load(open(golden_path("plonk_small.json")))
    rows, w = synth.make(meta["n"], meta["p"], meta["m"], meta["seed"])
    zk = pk.setup(meta["n"], meta["p"], rows, int(meta["tau"]))
    assert pk.write_zkey(zk) == open(golden_path("plonk_small.zkey"), "rb").read()
    assert f.write_wtns(w) == open(golden_path("plonk_small.wtns"), "rb").read()
    bl = {i + 1: int(x) for i, x in enumerate(meta["blinding"])}
    proof, pub = pk.prove(zk, w, bl)
    assert pk.proof_obj(proof) == meta["proof"] and [str(x) for x in pub] == meta["public"]
    assert pk.verify(pk.vkey(zk), pub, pk.proof_from_obj(meta["proof"]))

"""SURVEY 8f row 4, PLONK prover: the device prover (csrc/plonk.hip) through the C ABI against oracle/plonk.py on the
SAME zkey (written by the oracle in snarkjs's PLONK layout), witness and blinding scalars b1..b9 -- the proof must be
the oracle's proof bit for bit (every commitment, every evaluation), and the oracle's KZG verifier must accept it."""
import pytest

import bn254 as b
import formats as f
import plonk as pk
import synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n,p,m,seed", [(24, 2, 12, 1), (60, 5, 40, 3), (300, 20, 260, 9), (700, 513, 150, 4)])
def test_proof_equals_oracle(amd, n, p, m, seed):
    rows, w = synth.make(n, p, m, seed)
    zk = pk.setup(n, p, rows, tau=0xabcdef + seed)
    zkey = pk.write_zkey(zk, with_lagrange=(seed % 2 == 1))      # the prover does not read section 13
    wtns = f.write_wtns(w)
    rng = synth.Xoshiro(seed + 40)
    bl = {i: rng.rand_fr() for i in range(1, 10)}
    prover = amd.PlonkProver(zkey, device=0)
    assert (prover.n_public, prover.domain_size, prover.n_additions) == (p, zk["domainSize"], zk["nAdditions"])
    proof, pub = prover.prove(wtns, [bl[i] for i in range(1, 10)])
    exp, exp_pub = pk.prove(zk, w, bl)
    assert proof == pk.proof_obj(exp)
    assert pub == [str(x) for x in exp_pub]
    assert pk.verify(pk.vkey(zk), [int(x) for x in pub], pk.proof_from_obj(proof))
    # random blinding: another proof, accepted as well; a second witness of the same circuit
    proof2, pub2 = prover.prove(wtns)
    assert proof2["A"] != proof["A"] and pk.verify(pk.vkey(zk), [int(x) for x in pub2], pk.proof_from_obj(proof2))
    w3 = synth.make(n, p, m, seed, 777)[1]
    proof3, pub3 = prover.prove(f.write_wtns(w3), [bl[i] for i in range(1, 10)])
    assert proof3 == pk.proof_obj(pk.prove(zk, w3, bl)[0])
    prover.close()


def test_plonk_errors(amd):
    rows, w = synth.make(24, 2, 12, 1)
    zk = pk.setup(24, 2, rows, tau=99)
    zkey = pk.write_zkey(zk)
    with pytest.raises(amd.G16Error) as e:
        amd.PlonkProver(open(__import__("conftest").golden_path("tiny.zkey"), "rb").read())
    assert "zkey file is not plonk" in str(e.value)
    with pytest.raises(amd.G16Error) as e:
        amd.PlonkProver(zkey[:300])
    assert "Invalid File format" in str(e.value)
    prover = amd.PlonkProver(zkey)
    with pytest.raises(amd.G16Error) as e:
        prover.prove(f.write_wtns(w[:-1]))
    assert "Invalid witness length. Circuit: %d, witness: %d, %d" % (zk["nVars"], len(w) - 1, zk["nAdditions"]) in str(e.value)
    bad = list(w)
    bad[len(w) - 1] = (bad[len(w) - 1] + 1) % b.R           # breaks a gate: the quotient is no polynomial
    with pytest.raises(amd.G16Error) as e:
        prover.prove(f.write_wtns(bad))
    assert "not divisible" in str(e.value) or "Copy constraints" in str(e.value) or "does not divide" in str(e.value)
    prover.close()


@pytest.mark.parametrize("n,p,m,seed,lag", [(24, 2, 12, 1, True), (300, 20, 260, 9, True), (700, 513, 150, 4, False)])
def test_setup_tool_equals_oracle(amd, n, p, m, seed, lag):
    """g16_plonk_setup (R1CS -> gates on the host, transforms and powers of tau on the device) writes the zkey
    oracle/plonk.py::setup + write_zkey write for the same tau, byte for byte -- and that key proves."""
    import groth16 as g
    _, rows, _ = synth.gen_circuit(n, p, m, seed)
    r1cs = f.write_r1cs(n, p, 0, rows)
    zkey = amd.plonk_setup(r1cs, seed, device=0, with_lagrange=lag)
    rows_w, w = synth.make(n, p, m, seed)
    tau = g.trapdoor(seed + 1)["tau"]
    zk = pk.setup(n, p, rows_w, tau)
    assert zkey == pk.write_zkey(zk, with_lagrange=lag)
    prover = amd.PlonkProver(zkey, device=0)
    proof, pub = prover.prove(f.write_wtns(w))
    prover.close()
    assert pk.verify(pk.vkey(zk), [int(x) for x in pub], pk.proof_from_obj(proof))


def test_full_size_nzcp_example_as_plonk(amd):
    """The circuit whose PLONK setup the reference scripts (/root/reference/Makefile:30-33), at full size: natively built
    nzcp_exampleTest -> 2.59 M gates, domain 2^22 -> device setup (known tau), device proof, accepted by the oracle's
    KZG verifier; its public signals are the reference's 513 golden values of the example pass."""
    from test_cpu_sha256_circuit import example_public_signals, example_to_be_signed
    tbs = example_to_be_signed()
    out = amd.nzcp_circuit_setup(amd.NZCP_EXAMPLE_PARAMS, tbs, 7, 0, want_zkey=False, want_r1cs=True)
    zkey = amd.plonk_setup(out["r1cs"], 7, device=0, with_lagrange=False)
    vk = pk.vkey_from_zkey(zkey)
    prover = amd.PlonkProver(zkey, device=0)
    del zkey
    assert prover.domain_size == 1 << 22 and prover.n_public == 513
    proof, pub = prover.prove(out["wtns"])
    prover.close()
    assert pub == [str(x) for x in example_public_signals()]       # /root/reference/test/nzcp.js:41-49
    assert pk.verify(vk, [int(x) for x in pub], pk.proof_from_obj(proof))
    bad = list(pub)
    bad[100] = str(1 - int(bad[100]))
    assert not pk.verify(vk, [int(x) for x in bad], pk.proof_from_obj(proof))


def test_setup_edge_shapes_of_r1cs_rows(amd):
    """The branches of the R1CS -> PLONK conversion the random generator never takes: a constant times a linear
    combination (A or B has only the constant wire), an empty A (the row is C = 0), repeated wires inside one linear
    combination, long linear combinations (chains of addition gates), a row with constants on both sides of a product."""
    R = b.R
    n, p = 12, 2
    # wires: 0 = 1, 1..2 public, 3..11 private
    w = [1, 6, 35, 2, 3, 5, 7, 11, 13, 0, 0, 0]
    w[9] = (w[3] + 2 * w[4] + 3 * w[5] + 4 * w[6] + 5 * w[7] + 6 * w[8]) % R           # long sum
    w[10] = (w[9] + 7) * (w[3] + 1) % R                                                # constants on both factors
    w[11] = 4 * (w[4] + w[5]) % R
    rows = [
        ([(3, 1)], [(4, 1)], [(1, 1)]),                                                 # 2 * 3 = 6 (public)
        ([(5, 1)], [(6, 1)], [(2, 1)]),                                                 # 5 * 7 = 35 (public)
        ([(0, 1)], [(3, 1), (4, 2), (5, 3), (6, 4), (7, 5), (8, 6)], [(9, 1)]),          # A = constant 1
        ([(9, 1), (0, 7)], [(3, 1), (0, 1)], [(10, 1)]),                                # (w9 + 7)(w3 + 1) = w10
        ([(4, 1), (5, 1)], [(0, 4)], [(11, 1)]),                                        # B = constant 4
        ([], [(3, 1)], [(11, 1), (4, R - 4), (5, R - 4)]),                              # A empty: C = 0
        ([(3, 1), (3, 2), (4, 1)], [(5, 1), (5, 1)], [(3, 30), (4, 10)]),               # repeated wires: (3 w3 + w4)(2 w5) = 30 w3 + 10 w4
    ]
    assert synth.check_r1cs(rows, w) if hasattr(synth, "check_r1cs") else True
    gates, adds, pnv = pk.r1cs_to_plonk(n, p, rows)
    assert pk.check_gates(gates, p, pk.extend_witness(w, adds)) and len(adds) >= 5
    import groth16 as g
    seed = 21
    zk = pk.setup(n, p, rows, g.trapdoor(seed + 1)["tau"])
    zkey = amd.plonk_setup(f.write_r1cs(n, p, 0, rows), seed, device=0)
    assert zkey == pk.write_zkey(zk)
    prover = amd.PlonkProver(zkey)
    rng = synth.Xoshiro(5)
    bl = {i: rng.rand_fr() for i in range(1, 10)}
    proof, pub = prover.prove(f.write_wtns(w), [bl[i] for i in range(1, 10)])
    prover.close()
    assert proof == pk.proof_obj(pk.prove(zk, w, bl)[0]) and pub == ["6", "35"]
    assert pk.verify(pk.vkey(zk), [6, 35], pk.proof_from_obj(proof))


def test_setup_from_a_ptau_file_equals_oracle(amd):
    """`snarkjs plonk setup c.r1cs pot.ptau c.zkey` (/root/reference/Makefile:31) with the powers taken from a .ptau
    image (written by the oracle for a known tau): the zkey -- commitments by MSM over the file's points this time --
    equals the oracle's for that tau byte for byte; a circuit that does not fit the ceremony is refused with snarkjs's text."""
    n, p, m, seed = 300, 20, 260, 9
    _, rows, _ = synth.gen_circuit(n, p, m, seed)
    r1cs = f.write_r1cs(n, p, 0, rows)
    rows_w, w = synth.make(n, p, m, seed)
    tau = 0x5eed1234abcdef
    zk = pk.setup(n, p, rows_w, tau)
    ptau = pk.write_ptau(zk["power"] + 1, tau)          # a bigger ceremony than needed, as in practice
    zkey = amd.plonk_setup_ptau(r1cs, ptau, device=0)
    assert zkey == pk.write_zkey(zk)
    prover = amd.PlonkProver(zkey)
    proof, pub = prover.prove(f.write_wtns(w))
    prover.close()
    assert pk.verify(pk.vkey_from_zkey(zkey), [int(x) for x in pub], pk.proof_from_obj(proof))
    small = pk.write_ptau(zk["power"] - 1, tau)
    with pytest.raises(amd.G16Error) as e:
        amd.plonk_setup_ptau(r1cs, small, device=0)
    assert "circuit too big for this power of tau ceremony" in str(e.value) and "2**%d" % (zk["power"] - 1) in str(e.value)
    with pytest.raises(amd.G16Error) as e:
        amd.plonk_setup_ptau(r1cs, ptau[:500], device=0)
    assert "Invalid File format" in str(e.value)


def _vk_json(zk):
    """verification_key.json of a PLONK key, as snarkjs writes it (decimal strings)."""
    def g1(P):
        return ["0", "1", "0"] if P is None else [str(P[0]), str(P[1]), "1"]
    X2 = zk["X_2"]
    vk = {"protocol": "plonk", "curve": "bn128", "nPublic": zk["nPublic"], "power": zk["power"], "k1": str(zk["k1"]), "k2": str(zk["k2"])}
    for k in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
        vk[k] = g1(zk[k])
    vk["X_2"] = [[str(X2[0][0]), str(X2[0][1])], [str(X2[1][0]), str(X2[1][1])], ["1", "0"]]
    vk["w"] = str(b.fr_root(zk["power"]))
    return vk


@pytest.mark.parametrize("n,p,m,seed", [(60, 5, 40, 3), (700, 513, 150, 4)])
def test_plonk_verifier_verdicts_equal_oracle(amd, n, p, m, seed):
    """`snarkjs plonk verify` on the device (csrc/verify_plonk.hip): every verdict -- valid proofs, a wrong public
    signal, tampered evaluations and commitments, a point off the curve, a non-canonical coordinate, too few public
    signals -- equals oracle/plonk.py::verify's."""
    import copy
    rows, w = synth.make(n, p, m, seed)
    zk = pk.setup(n, p, rows, tau=0xfeed + seed)
    vk = pk.vkey(zk)
    rng = synth.Xoshiro(seed + 3)
    proofs = []
    for k in range(3):
        wk = w if k == 0 else synth.make(n, p, m, seed, 500 + k)[1]
        pr, pub = pk.prove(zk, wk, {i: rng.rand_fr() for i in range(1, 10)})
        proofs.append((pub, pr))
    cases = [([str(x) for x in pub], pk.proof_obj(pr)) for pub, pr in proofs]
    pub0, pr0 = cases[0]
    bad = list(pub0)
    bad[p - 1] = str((int(bad[p - 1]) + 1) % b.R)
    cases.append((bad, pr0))
    for key in ("eval_a", "eval_s2", "eval_zw", "eval_r"):
        t = copy.deepcopy(pr0)
        t[key] = str((int(t[key]) + 1) % b.R)
        cases.append((pub0, t))
    for key in ("A", "Z", "T2", "Wxi", "Wxiw"):
        t = copy.deepcopy(pr0)
        t[key] = list(cases[1][1][key])           # a valid point of another proof
        cases.append((pub0, t))
    t = copy.deepcopy(pr0)
    t["T1"][0] = str((int(t["T1"][0]) + 1) % b.Q)  # off the curve
    cases.append((pub0, t))
    t = copy.deepcopy(pr0)
    t["eval_b"] = str(int(t["eval_b"]) + b.R)      # evaluation given as value + r: reduced, still valid
    if int(t["eval_b"]) < (1 << 256):
        cases.append((pub0, t))
    ver = amd.PlonkVerifier(_vk_json(zk))
    got = ver.verify_batch(cases)

    def oracle(ps, po):
        try:
            return pk.verify(vk, [int(x) % b.R for x in ps], pk.proof_from_obj(po))
        except Exception:
            return False
    exp = [oracle(ps, po) for ps, po in cases]
    assert got == exp
    assert got[:3] == [True, True, True] and not any(got[3:13])
    assert ver.verify(pub0, pr0) is True and ver.verify(pub0[:-1], pr0) is False
    t = copy.deepcopy(pr0)
    t["C"][1] = str(int(t["C"][1]) + b.Q)          # non-canonical coordinate: rejected (the oracle would reduce it)
    assert ver.verify(pub0, t) is False
    # the device prover's proof through the device verifier
    prover = amd.PlonkProver(pk.write_zkey(zk))
    gp, gpub = prover.prove(f.write_wtns(w))
    prover.close()
    assert ver.verify(gpub, gp) is True
    ver.close()


def test_device_prover_and_verifier_on_the_golden_fixture(amd):
    """tests/golden/plonk_small.*: the committed key and witness, proved on the device with the committed blinding, give
    the committed proof.json / public.json; the device verifier accepts it against the committed verification key."""
    import json
    from conftest import golden_path
    meta = json.load(open(golden_path("plonk_small.json")))
    prover = amd.PlonkProver(open(golden_path("plonk_small.zkey"), "rb").read())
    proof, pub = prover.prove(open(golden_path("plonk_small.wtns"), "rb").read(), [int(x) for x in meta["blinding"]])
    prover.close()
    assert proof == meta["proof"] and pub == meta["public"]
    ver = amd.PlonkVerifier(meta["vkey"])
    assert ver.verify(pub, proof) is True
    assert ver.verify([str(int(pub[0]) + 1)] + pub[1:], proof) is False
    ver.close()

"""GPU parity at BASELINE.json's FULL sizes, where the big-int oracle cannot
follow.  Checks are the size-independent ones the domain offers:

  * the proof is accepted by the Groth16 pairing equation against the setup's verification key
    (e(-A,B) e(alpha,beta) e(vk_x,gamma) e(C,delta) == 1, oracle/bn254.py),
  * bit-equality with the oracle's C prover (oracle/c) on the SAME (zkey, wtns, r, s),
  * public.json == w[1..p] of the witness,
  * linearity of the blinding: proofs of one witness under (r, s) and (r, s + d) have
    pi_b' = pi_b + d.delta2, and both verify,
  * determinism: a second run gives the same bytes; batch == one-by-one; point-range shards == whole.
Inputs are shape-matched synthetic circuits (the real circuit cannot be compiled offline)."""
import ctypes
import hashlib
import os

import pytest

import bn254 as b
import formats as f
import groth16 as g
import synth
from conftest import ROOT

pytestmark = pytest.mark.gpu

P_NZCP = 513
WTNS_BODY = 12 + (12 + 40) + 12     # binfile header, section 1 (n8, r, nWitness), section 2 header


def _oracle_c():
    path = os.path.join(ROOT, "oracle", "_build", "libg16oracle.so")
    # the checker must be there under -m gpu: a missing oracle is a failure, not a skip (__graft_entry__.build() makes it)
    assert os.path.exists(path), "oracle/_build/libg16oracle.so not built: run __graft_entry__.build()"
    lib = ctypes.CDLL(path)
    lib.g16o_prove.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                               ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    return lib


def _rs(seed):
    rng = synth.Xoshiro(seed + 2)
    return rng.rand_fr(), rng.rand_fr()


def _vk(vkey, p=P_NZCP):
    return {"alpha1": f.g1_from_lem(vkey[0:64]), "beta2": f.g2_from_lem(vkey[64:192]),
            "gamma2": f.g2_from_lem(vkey[192:320]), "delta2": f.g2_from_lem(vkey[320:448]),
            "IC": [f.g1_from_lem(vkey[448 + 64 * i:512 + 64 * i]) for i in range(p + 1)]}


def _points(proof):
    return f.g1_from_obj(proof["pi_a"]), f.g2_from_obj(proof["pi_b"]), f.g1_from_obj(proof["pi_c"])


def _pub_of(wtns, p=P_NZCP):
    return [str(f.from_le(wtns[WTNS_BODY + 32 * i:WTNS_BODY + 32 * (i + 1)])) for i in range(1, p + 1)]


def _verifies(vk, pub, proof):
    return g.verify(vk, [int(x) for x in pub], _points(proof))


@pytest.mark.parametrize("n", [900_000, 1_700_000])
def test_config2_nzcp_live_shape(amd, n):
    """config 2: n = nConstraints in {0.9 M, 1.7 M}, p = 513."""
    seed = synth.SEED_NZCP
    zkey, wtns, vkey = amd.synth_setup(n, P_NZCP, n, seed)
    vk = _vk(vkey)
    r, s = _rs(seed)
    prover = amd.Prover(zkey)
    assert prover.info.domain_size == (1 << 20 if n == 900_000 else 1 << 21)
    proof, pub = prover.prove(wtns, f.le(r), f.le(s))
    assert pub == _pub_of(wtns)
    assert _verifies(vk, pub, proof)
    # same bytes from the oracle's C prover
    olib = _oracle_c()
    out = ctypes.create_string_buffer(256)
    opub = ctypes.create_string_buffer(P_NZCP * 32)
    assert olib.g16o_prove(zkey, len(zkey), wtns, len(wtns), f.le(r), f.le(s), out, opub, os.cpu_count() or 1) == 0
    pr = amd.Proof()
    gpub = ctypes.create_string_buffer(P_NZCP * 32)
    prover.stage(0, wtns)
    assert prover.prove_staged_raw(0, f.le(r), f.le(s), pr, gpub) == 0
    assert bytes(pr.a) + bytes(pr.b) + bytes(pr.c) == out.raw
    assert gpub.raw == opub.raw
    # determinism + blinding linearity
    proof2, _ = prover.prove(wtns, f.le(r), f.le(s))
    assert proof2 == proof
    d = 99
    proof3, _ = prover.prove(wtns, f.le(r), f.le((s + d) % b.R))
    assert proof3["pi_a"] == proof["pi_a"]
    assert b.G2.add(_points(proof)[1], b.G2.mul(vk["delta2"], d)) == _points(proof3)[1]
    assert _verifies(vk, pub, proof3)
    prover.close()


def test_config3_batch_throughput_mode(amd):
    """config 3 (bounded): one resident key, a batch of independent witnesses through g16_prove_batch;
    every proof equals the one-by-one result, first and last are pairing-verified."""
    n, seed, count = 900_000, synth.SEED_NZCP, 6
    zkey, wtns0, vkey = amd.synth_setup(n, P_NZCP, n, seed)
    vk = _vk(vkey)
    wts = [wtns0] + [amd.synth_witness(n, P_NZCP, n, seed, seed + 100 + i) for i in range(1, count)]
    assert len({hashlib.sha256(w).digest() for w in wts}) == count
    lib = amd.load()
    prover = amd.Prover(zkey)
    rs = [_rs(seed + i) for i in range(count)]
    arr = (ctypes.c_char_p * count)(*wts)
    lens = (ctypes.c_size_t * count)(*[len(w) for w in wts])
    rsbuf = b"".join(f.le(r) + f.le(s) for r, s in rs)
    proofs = (amd.Proof * count)()
    pubs = ctypes.create_string_buffer(count * P_NZCP * 32)
    rc = lib.g16_prove_batch(prover._h, arr, lens, count, rsbuf, proofs, pubs)
    assert rc == 0, lib.g16_last_error()
    for i in range(count):
        obj = amd.proof_to_obj(proofs[i])
        raw = pubs.raw[i * P_NZCP * 32:(i + 1) * P_NZCP * 32]
        pub = [str(f.from_le(raw[k * 32:(k + 1) * 32])) for k in range(P_NZCP)]
        assert pub == _pub_of(wts[i])
        one, _ = prover.prove(wts[i], f.le(rs[i][0]), f.le(rs[i][1]))
        assert obj == one
        if i in (0, count - 1):
            assert _verifies(vk, pub, obj)
    prover.close()


def test_config3_batch_nominal(amd):
    """config 3 AS STATED in BASELINE.json: a batch of 1 024 independent nzcp_live proofs (n = 1.7 M, the upper
    structural estimate) through ONE g16_prove_batch call, cycling 8 distinct witnesses with a fresh blinding pair
    for every proof.  Every proof must equal its one-by-one result (same witness, same (r, s)); the first, the
    last and a random one are pairing-verified against the setup's verification key."""
    n, seed, count, distinct = 1_700_000, synth.SEED_NZCP, 1024, 8
    zkey, wtns0, vkey = amd.synth_setup(n, P_NZCP, n, seed)
    vk = _vk(vkey)
    wts = [wtns0] + [amd.synth_witness(n, P_NZCP, n, seed, seed + 200 + i) for i in range(1, distinct)]
    assert len({hashlib.sha256(w).digest() for w in wts}) == distinct
    lib = amd.load()
    prover = amd.Prover(zkey)
    del zkey
    # the blinding pairs cycle with period 9 (coprime to 8): 72 distinct (witness, r, s) triples, each checked once
    # against a one-by-one proof, every batch proof compared with the triple it repeats
    rs = [_rs(seed + 1000 + i) for i in range(9)]
    arr = (ctypes.c_char_p * count)(*[wts[i % distinct] for i in range(count)])
    lens = (ctypes.c_size_t * count)(*[len(wts[i % distinct]) for i in range(count)])
    rsbuf = b"".join(f.le(rs[i % 9][0]) + f.le(rs[i % 9][1]) for i in range(count))
    proofs = (amd.Proof * count)()
    pubs = ctypes.create_string_buffer(count * P_NZCP * 32)
    rc = lib.g16_prove_batch(prover._h, arr, lens, count, rsbuf, proofs, pubs)
    assert rc == 0, lib.g16_last_error()
    ref = {}
    pr, gpub = amd.Proof(), ctypes.create_string_buffer(P_NZCP * 32)
    for k in range(72):
        wi, ri = k % distinct, k % 9
        prover.stage(0, wts[wi])
        assert prover.prove_staged_raw(0, f.le(rs[ri][0]), f.le(rs[ri][1]), pr, gpub) == 0
        ref[(wi, ri)] = (bytes(pr.a) + bytes(pr.b) + bytes(pr.c), gpub.raw)
    assert len({v[0] for v in ref.values()}) == 72
    for i in range(count):
        want, wpub = ref[(i % distinct, i % 9)]
        assert bytes(proofs[i].a) + bytes(proofs[i].b) + bytes(proofs[i].c) == want, i
        assert pubs.raw[i * P_NZCP * 32:(i + 1) * P_NZCP * 32] == wpub, i
    for i in (0, count - 1, 517):
        raw = pubs.raw[i * P_NZCP * 32:(i + 1) * P_NZCP * 32]
        pub = [str(f.from_le(raw[k * 32:(k + 1) * 32])) for k in range(P_NZCP)]
        assert pub == _pub_of(wts[i % distinct])
        assert _verifies(vk, pub, amd.proof_to_obj(proofs[i]))
    prover.close()


@pytest.mark.parametrize("which", ["example", "live"])
def test_real_nzcp_circuit_with_in_circuit_cbor_search(amd, which):
    """The REAL constraint system of BASELINE configs 1 / 2, built natively (csrc/nzcp_gadgets.h): NZCPPubIdentity
    with the CBOR search in the circuit -- nzcp_exampleTest.circom on the MoH example pass, nzcp_liveTest.circom on a
    live-format pass.  GPU proof: public.json equals the reference's golden public signals (example) / the values
    its test computes from the pass (live: SHA-256 of "given,family,dob", SHA-256 of ToBeSigned, exp:
    /root/reference/test/nzcp.js:18-49), the proof is accepted by the pairing check against the setup's
    verification key and equals the CPU oracle's bytes for the same (zkey, wtns, r, s)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import nzcp_pass
    from test_cpu_sha256_circuit import example_public_signals, example_to_be_signed
    if which == "example":
        tbs, params = example_to_be_signed(), amd.NZCP_EXAMPLE_PARAMS
        want = [str(x) for x in example_public_signals()]
    else:
        tbs, params = nzcp_pass.to_be_signed("Anne-Marie", "Te Whare", "1987-11-30", live=True, exp=1700000000), amd.NZCP_LIVE_PARAMS
        want = [str(x) for x in _bits_msb_first(hashlib.sha256(b"Anne-Marie,Te Whare,1987-11-30").digest())] + \
            [str(x) for x in _bits_msb_first(hashlib.sha256(tbs).digest())] + ["1700000000"]
    out = amd.nzcp_circuit_setup(params, tbs, 77)
    assert 600_000 < out["n_constraints"] < 1_000_000
    vk = _vk(out["vkey"])
    r, s = _rs(77)
    prover = amd.Prover(out["zkey"])
    assert prover.info.domain_size == 1 << 20 and prover.info.n_public == P_NZCP
    proof, pub = prover.prove(out["wtns"], f.le(r), f.le(s))
    assert pub == want
    assert _verifies(vk, pub, proof)
    olib = _oracle_c()
    obuf = ctypes.create_string_buffer(256)
    opub = ctypes.create_string_buffer(P_NZCP * 32)
    assert olib.g16o_prove(out["zkey"], len(out["zkey"]), out["wtns"], len(out["wtns"]), f.le(r), f.le(s), obuf, opub,
                           os.cpu_count() or 1) == 0
    pr = amd.Proof()
    gpub = ctypes.create_string_buffer(P_NZCP * 32)
    prover.stage(0, out["wtns"])
    assert prover.prove_staged_raw(0, f.le(r), f.le(s), pr, gpub) == 0
    assert bytes(pr.a) + bytes(pr.b) + bytes(pr.c) == obuf.raw and gpub.raw == opub.raw
    prover.close()


def _bits_msb_first(data):
    return [str((byte >> (7 - k)) & 1) for byte in data for k in range(8)]


@pytest.mark.parametrize("blocks", [2, 163])
def test_config5_sha256_chain_real_circuit(amd, blocks):
    """config 5 with a REAL constraint system: `blocks` chained SHA-256 compressions (163 fill the 2^22 domain),
    witness of bits only.  public.json must be the digest bits hashlib computes; the proof must verify; at the
    small size it must also equal the C oracle's bytes."""
    msg = hashlib.sha256(b"nzcp-circom config 5").digest()
    seed = synth.SEED_NZCP + 50 + blocks
    out = amd.sha256_chain_setup(blocks, msg, seed)
    zkey, wtns, vk = out["zkey"], out["wtns"], _vk(out["vkey"], 256)
    d = msg
    for _ in range(blocks):
        d = hashlib.sha256(d).digest()
    r, s = _rs(seed)
    prover = amd.Prover(zkey)
    assert prover.info.n_public == 256
    assert prover.info.domain_size == (1 << 22 if blocks == 163 else 1 << 16)
    proof, pub = prover.prove(wtns, f.le(r), f.le(s))
    assert pub == _bits_msb_first(d)
    assert g.verify(vk, [int(x) for x in pub], _points(proof))
    if blocks == 2:
        olib = _oracle_c()
        obuf = ctypes.create_string_buffer(256)
        opub = ctypes.create_string_buffer(256 * 32)
        assert olib.g16o_prove(zkey, len(zkey), wtns, len(wtns), f.le(r), f.le(s), obuf, opub, os.cpu_count() or 1) == 0
        pr = amd.Proof()
        gpub = ctypes.create_string_buffer(256 * 32)
        prover.stage(0, wtns)
        assert prover.prove_staged_raw(0, f.le(r), f.le(s), pr, gpub) == 0
        assert bytes(pr.a) + bytes(pr.b) + bytes(pr.c) == obuf.raw and gpub.raw == opub.raw
    prover.close()


def test_config5_stress_2_22(amd):
    """config 5: N = 2^22 (n = nConstraints = 2^22 - 514), the largest configuration in BASELINE.json."""
    n = (1 << 22) - (P_NZCP + 1)
    seed = synth.SEED_NZCP + 5
    zkey, wtns, vkey = amd.synth_setup(n, P_NZCP, n, seed)
    vk = _vk(vkey)
    r, s = _rs(seed)
    prover = amd.Prover(zkey)
    assert prover.info.domain_size == 1 << 22
    proof, pub = prover.prove(wtns, f.le(r), f.le(s))
    assert pub == _pub_of(wtns)
    assert _verifies(vk, pub, proof)
    prover.close()
    # config 4's sharding at this size: 3 uneven point-range shards on the same GPU == the whole proof
    parts = []
    for k in range(3):
        sh = amd.Prover(zkey, shard_rank=k, shard_count=3)
        sh.stage(0, wtns)
        parts.append(sh.prove_partial(0))
        sh.close()
    assert amd.finish_host(zkey, parts, f.le(r), f.le(s)) == proof


def test_config4_eight_shards_on_the_live_circuit(amd):
    """BASELINE config 4 on its own circuit: the nzcp_liveTest constraint system sharded EIGHT ways through the one-process
    host (g16_multi_*: what Node's createProver(zkey, {devices}) binds) -- all eight shard handles on this GPU, the witness
    uploaded once and fanned out, slices of the A/B/C coset evaluations moved by asynchronous device copies behind events.
    A shard of 2^17 H points picks its own window size (the narrow-top-window cliff of r02 sat exactly here).  The proof
    must equal the unsharded handle's bytes, twice (the second proof reuses every buffer and thread)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import nzcp_pass
    tbs = nzcp_pass.to_be_signed("Aroha", "Ngata", "1975-07-21", live=True, exp=1800000000)
    out = amd.nzcp_circuit_setup(amd.NZCP_LIVE_PARAMS, tbs, 78)
    r, s = _rs(78)
    prover = amd.Prover(out["zkey"])
    proof, pub = prover.prove(out["wtns"], f.le(r), f.le(s))
    prover.close()
    assert _verifies(_vk(out["vkey"]), pub, proof)
    mp = amd.MultiProver(out["zkey"], [0] * 8)
    assert mp.n_shards == 8
    for _ in range(2):
        p8, pub8 = mp.prove(out["wtns"], f.le(r), f.le(s))
        assert p8 == proof and pub8 == pub
    # a witness word that is not reduced modulo r is refused (the check runs on shard 0, behind the upload)
    bad = bytearray(out["wtns"])
    bad[WTNS_BODY + 32 * 5:WTNS_BODY + 32 * 6] = b"\xff" * 32
    with pytest.raises(amd.G16Error, match="not reduced"):
        mp.prove(bytes(bad), f.le(r), f.le(s))
    p8, _ = mp.prove(out["wtns"], f.le(r), f.le(s))
    assert p8 == proof
    mp.close()

"""GPU parity at the drop-in boundary: g16_create/g16_prove through the C ABI against
(1) the committed golden fixtures (tests/golden, produced by the Python oracle and pinned by the
trapdoor KAT + pairing check), (2) the Python oracle on fresh seeded circuits, (3) the trapdoor
known-answer at sizes the big-int oracle cannot reach.  Mirrors what a snarkjs user would test:
`groth16.prove(zkey, wtns)` -> proof.json / public.json, then `groth16.verify`."""
import json
import struct

import pytest

import bn254 as b
import formats as f
import groth16 as g
import synth
from conftest import golden_path

pytestmark = pytest.mark.gpu


def _golden(name):
    zk = open(golden_path(name + ".zkey"), "rb").read()
    wt = open(golden_path(name + ".wtns"), "rb").read()
    meta = json.load(open(golden_path(name + ".json")))
    return zk, wt, meta


@pytest.mark.parametrize("name", ["tiny", "small", "nzcp513"])
def test_golden_proof_bytes(amd, name):
    zk, wt, meta = _golden(name)
    prover = amd.Prover(zk)
    proof, pub = prover.prove(wt, f.le(int(meta["r"])), f.le(int(meta["s"])))
    assert pub == meta["public"]
    assert proof == meta["proof"]
    # byte-identical proof.json / public.json (JSON.stringify(x, null, 1))
    assert amd.stringify(proof) == f.js_stringify(meta["proof"])
    assert amd.stringify(pub) == f.js_stringify(meta["public"])
    prover.close()


@pytest.mark.parametrize("n,p,m,seed,c", [(40, 1, 30, 21, 0), (300, 4, 250, 22, 5), (1200, 9, 1000, 23, 0),
                                          (900, 513, 380, 24, 7)])
def test_prove_matches_python_oracle(amd, n, p, m, seed, c):
    zkey, wtns, _ = amd.synth_setup(n, p, m, seed)
    zk = f.read_zkey(zkey)
    w = f.read_wtns(wtns)["w"]
    rng = synth.Xoshiro(seed + 2)
    r, s = rng.rand_fr(), rng.rand_fr()
    prover = amd.Prover(zkey, window_bits=c)
    proof, pub = prover.prove(wtns, f.le(r), f.le(s))
    (A, B, C), opub = g.prove(zk, w, r, s)
    assert proof == f.proof_obj(A, B, C)
    assert pub == [str(x) for x in opub]
    assert g.verify(zk, opub, (A, B, C))
    # a second, independent witness against the same resident key (batch mode, config 3)
    wt2 = amd.synth_witness(n, p, m, seed, seed + 1000)
    proof2, pub2 = prover.prove(wt2, f.le(r), f.le(s))
    w2 = f.read_wtns(wt2)["w"]
    (A2, B2, C2), opub2 = g.prove(zk, w2, r, s)
    assert proof2 == f.proof_obj(A2, B2, C2) and pub2 == [str(x) for x in opub2]
    assert g.verify(zk, opub2, (A2, B2, C2))
    prover.close()


def test_random_blinding_verifies(amd):
    """r = s = NULL -> CSPRNG blinding (snarkjs Fr.random()): proofs differ, both verify."""
    zk, wt, meta = _golden("small")
    zko = f.read_zkey(zk)
    prover = amd.Prover(zk)
    p1, pub1 = prover.prove(wt)
    p2, pub2 = prover.prove(wt)
    assert p1 != p2 and pub1 == pub2 == meta["public"]
    for pr in (p1, p2):
        pts = (f.g1_from_obj(pr["pi_a"]), f.g2_from_obj(pr["pi_b"]), f.g1_from_obj(pr["pi_c"]))
        assert g.verify(zko, [int(x) for x in pub1], pts)
    prover.close()


def test_sharded_partials_equal_single(amd):
    """Point-range sharding on one GPU: 3 shard handles -> partials -> finish == 1 GPU."""
    zk, wt, meta = _golden("nzcp513")
    r, s = f.le(int(meta["r"])), f.le(int(meta["s"]))
    parts = []
    last = None
    for rank in range(3):
        pv = amd.Prover(zk, shard_rank=rank, shard_count=3)
        pv.stage(0, wt)
        parts.append(pv.prove_partial(0))
        if last:
            last.close()
        last = pv
    proof, pub = last.prove_finish(0, parts, r, s)
    assert proof == meta["proof"] and pub == meta["public"]
    last.close()


@pytest.mark.parametrize("shards", [2, 3, 5])
def test_sharded_h_pipeline_equals_single(amd, shards):
    """BASELINE config 4 without the replicated NTT chain (g16_shard_begin / g16_shard_end): shard v mod G
    evaluates vector v of (A, B, C) on the coset, every shard receives its slice of all three, joins and
    multi-exponentiates only its own range.  Same proof bytes as one unsharded handle.  Exchange buffers are host
    memory here (between processes: an RCCL scatter; inside one process: peer copies)."""
    import ctypes as C
    zk, wt, meta = _golden("nzcp513")
    r, s = f.le(int(meta["r"])), f.le(int(meta["s"]))
    pvs = [amd.Prover(zk, shard_rank=k, shard_count=shards) for k in range(shards)]
    n_dom = pvs[0].info.domain_size
    eb = amd.LAZY_FR_BYTES
    vecs = [C.create_string_buffer(n_dom * eb) for _ in range(3)]
    for k, pv in enumerate(pvs):
        pv.stage(0, wt)
        mask = sum(1 << v for v in range(3) if v % shards == k)
        pv.shard_begin(0, mask, [C.addressof(vecs[v]) if (mask >> v) & 1 else 0 for v in range(3)])
    parts = []
    for k, pv in enumerate(pvs):
        lo, hi = amd.shard_range(n_dom, k, shards)
        sl = [C.create_string_buffer(vecs[v].raw[lo * eb:hi * eb], max(1, (hi - lo) * eb)) for v in range(3)]
        parts.append(pv.shard_end(0, [C.addressof(x) for x in sl]))
    proof, pub = pvs[-1].prove_finish(0, parts, r, s)
    assert proof == meta["proof"] and pub == meta["public"]
    # protocol errors are reported, not silently mis-proved
    with pytest.raises(amd.G16Error):
        pvs[0].shard_end(0, [0, 0, 0])                      # no begin
    with pytest.raises(amd.G16Error):
        pvs[0].prove_finish(0, parts[:-1], r, s)            # a missing partial
    for pv in pvs:
        pv.close()


@pytest.mark.parametrize("devices", [[0, 0], [0, 0, 0, 0]])
def test_multi_device_handle_equals_single(amd, devices):
    """g16_multi_*: one process, one sharded handle per listed device (here all on GPU 0), host threads per shard,
    slices moved with device copies, partial sums added on the host -- the C ABI under the Node host's
    createProver(zkey, {devices: [..]})."""
    zk, wt, meta = _golden("nzcp513")
    r, s = f.le(int(meta["r"])), f.le(int(meta["s"]))
    mp = amd.MultiProver(zk, devices)
    assert mp.n_shards == len(devices)
    for _ in range(2):
        proof, pub = mp.prove(wt, r, s)
        assert proof == meta["proof"] and pub == meta["public"]
    with pytest.raises(amd.G16Error, match="Invalid witness length"):
        mp.prove(_golden("tiny")[1], r, s)
    proof, pub = mp.prove(wt, r, s)        # still usable after a refused witness
    assert proof == meta["proof"]
    mp.close()


@pytest.mark.parametrize("nctx", [None, "1", "2"])
def test_batch_api(amd, nctx, monkeypatch):
    """g16_prove_batch pipelines over three proof contexts (G16_BATCH_CTX: sweeps); more proofs than contexts, so that
    every context is reused; each proof == the oracle's for its own witness."""
    import ctypes as C
    if nctx:
        monkeypatch.setenv("G16_BATCH_CTX", nctx)    # read at create
    zk, wt, meta = _golden("tiny")
    n, p, m, seed = meta["n"], meta["p"], meta["m"], meta["seed"]
    zko = f.read_zkey(zk)
    prover = amd.Prover(zk)
    wts = [wt] + [amd.synth_witness(n, p, m, seed, 500 + i) for i in range(6)]
    arr = (C.c_char_p * len(wts))(*wts)
    lens = (C.c_size_t * len(wts))(*[len(x) for x in wts])
    r, s = int(meta["r"]), int(meta["s"])
    rs = b"".join(f.le(r) + f.le(s) for _ in wts)
    out = (amd.Proof * len(wts))()
    pub = C.create_string_buffer(len(wts) * p * 32)
    rc = amd.load().g16_prove_batch(prover._h, arr, lens, len(wts), rs, out, pub)
    assert rc == 0, amd.load().g16_last_error()
    assert amd.proof_to_obj(out[0]) == meta["proof"]
    for i, wb in enumerate(wts):
        w = f.read_wtns(wb)["w"]
        (A, B, Cc), opub = g.prove(zko, w, r, s)
        assert amd.proof_to_obj(out[i]) == f.proof_obj(A, B, Cc)
        assert [int.from_bytes(pub.raw[(i * p + k) * 32:(i * p + k + 1) * 32], "little") for k in range(p)] == opub
    prover.close()


def test_prove_errors_match_snarkjs(amd):
    zk, wt, meta = _golden("tiny")
    prover = amd.Prover(zk)
    # witness of another circuit size
    other = amd.synth_witness(30, 2, 12, 1, 1)
    with pytest.raises(amd.G16Error, match=r"Invalid witness length. Circuit: 24, witness: 30"):
        prover.prove(other)
    # witness over another prime
    bad = bytearray(wt)
    secs = f.read_binfile(wt, "wtns", 2)
    pos, _ = secs[1][0]
    bad[pos + 4] ^= 1
    with pytest.raises(amd.G16Error, match="Curve of the witness does not match the curve of the proving key"):
        prover.prove(bytes(bad))
    with pytest.raises(amd.G16Error, match="wtns: Invalid File format"):
        prover.prove(b"zkey" + wt[4:])
    with pytest.raises(amd.G16Error, match="not reduced"):
        prover.prove(wt, f.le(b.R), f.le(1))
    # a witness word >= r is rejected (the recoding assumes canonical scalars)
    pos2, _ = secs[2][0]
    bad = bytearray(wt)
    bad[pos2 + 32 * 5:pos2 + 32 * 6] = f.le(b.R + 3)
    with pytest.raises(amd.G16Error, match="signal 5 is not reduced"):
        prover.prove(bytes(bad))
    prover.close()


def test_trapdoor_kat_medium(amd):
    """2^15-domain circuit: too slow for the big-int prover, so pin against the trapdoor known
    answer: proof == ([a]G1, [b]G2, [c]G1) with a, b, c from Fr arithmetic only."""
    n, p, m, seed = 30000, 513, 30000, 77
    zkey, wtns, _ = amd.synth_setup(n, p, m, seed)
    rows, w = synth.make(n, p, m, seed)
    assert f.write_wtns(w) == wtns
    # scalar side of the setup only (no point multiplications): u, v, t
    td = g.trapdoor(seed + 1)
    N = 1 << 15
    L = g.lagrange_at(N, td["tau"])
    u = [0] * n; v = [0] * n; t = [0] * n
    for c, (A, B, C) in enumerate(rows):
        for sg, cf in A: u[sg] = (u[sg] + cf * L[c]) % b.R
        for sg, cf in B: v[sg] = (v[sg] + cf * L[c]) % b.R
        for sg, cf in C: t[sg] = (t[sg] + cf * L[c]) % b.R
    for i in range(p + 1):
        u[i] = (u[i] + L[m + i]) % b.R
    sec = {"u": u, "v": v, "t": t, **td}
    rng = synth.Xoshiro(seed + 2)
    r, s = rng.rand_fr(), rng.rand_fr()
    exp = g.expected_proof(sec, p, w, r, s)
    prover = amd.Prover(zkey)
    assert prover.info.domain_size == N
    proof, pub = prover.prove(wtns, f.le(r), f.le(s))
    assert proof == f.proof_obj(*exp)
    assert pub == [str(x) for x in w[1:p + 1]]
    prover.close()


def test_public_json_matches_reference_golden_tobesigned_hash(amd):
    """A REAL circuit proved on the GPU, pinned by the reference's own golden data: SHA-256 over the MoH example
    pass's ToBeSigned (314 bytes, 6 compressions, ~155 k constraints).  public.json must be the bits of
    271ce33d...f3ee -- the `toBeSignedSha256` slice of the NZCP circuit's public signals
    (/root/reference/test/nzcp.js:41-47, value from SURVEY App. D.2) -- and the proof must verify and equal
    the C oracle's bytes."""
    import ctypes
    import os
    from test_cpu_sha256_circuit import EXAMPLE_TBS_SHA256, example_to_be_signed
    tbs = example_to_be_signed()
    out = amd.sha256_message_setup(tbs, 4242)
    zkey, wtns = out["zkey"], out["wtns"]
    vkey = out["vkey"]
    vk = {"alpha1": f.g1_from_lem(vkey[0:64]), "beta2": f.g2_from_lem(vkey[64:192]),
          "gamma2": f.g2_from_lem(vkey[192:320]), "delta2": f.g2_from_lem(vkey[320:448]),
          "IC": [f.g1_from_lem(vkey[448 + 64 * i:512 + 64 * i]) for i in range(257)]}
    rng = synth.Xoshiro(4244)
    r, s = rng.rand_fr(), rng.rand_fr()
    prover = amd.Prover(zkey)
    assert prover.info.n_public == 256 and prover.info.domain_size == 1 << 18
    proof, pub = prover.prove(wtns, f.le(r), f.le(s))
    want = [str((byte >> (7 - k)) & 1) for byte in bytes.fromhex(EXAMPLE_TBS_SHA256) for k in range(8)]
    assert pub == want
    pts = (f.g1_from_obj(proof["pi_a"]), f.g2_from_obj(proof["pi_b"]), f.g1_from_obj(proof["pi_c"]))
    assert g.verify(vk, [int(x) for x in pub], pts)
    # a proof for a different message must not verify against these public signals
    bad = amd.sha256_message_setup(tbs[:-1] + bytes([tbs[-1] ^ 1]), 4242, want_zkey=False)["wtns"]
    proof2, pub2 = prover.prove(bad, f.le(r), f.le(s))
    assert pub2 != want
    pts2 = (f.g1_from_obj(proof2["pi_a"]), f.g2_from_obj(proof2["pi_b"]), f.g1_from_obj(proof2["pi_c"]))
    assert not g.verify(vk, [int(x) for x in want], pts2)
    assert g.verify(vk, [int(x) for x in pub2], pts2)
    lib_path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "_build", "libg16oracle.so")
    if os.path.exists(lib_path):
        olib = ctypes.CDLL(lib_path)
        olib.g16o_prove.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                    ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        obuf, opub = ctypes.create_string_buffer(256), ctypes.create_string_buffer(256 * 32)
        assert olib.g16o_prove(zkey, len(zkey), wtns, len(wtns), f.le(r), f.le(s), obuf, opub, os.cpu_count() or 1) == 0
        assert f.proof_obj(f.g1_from_obj(proof["pi_a"]), f.g2_from_obj(proof["pi_b"]), f.g1_from_obj(proof["pi_c"])) == proof
        A = (f.from_le(obuf.raw[0:32]), f.from_le(obuf.raw[32:64]))
        assert proof["pi_a"][:2] == [str(A[0]), str(A[1])]
    prover.close()


def test_public_json_equals_reference_example_pass(amd):
    """public.json of a GPU proof == the 513 values the reference's test expects for the MoH example pass
    (/root/reference/test/nzcp.js:41-49; values SURVEY App. D.2), from the fixed-layout NZCP interface circuit
    over the pass's ToBeSigned; proof pairing-verified against the setup's verification key."""
    from test_cpu_sha256_circuit import (EXAMPLE_EXP_OFF, EXAMPLE_SEGS, example_public_signals, example_to_be_signed)
    out = amd.nzcp_fixed_layout_setup(example_to_be_signed(), EXAMPLE_SEGS, EXAMPLE_EXP_OFF, 777)
    vkey = out["vkey"]
    vk = {"alpha1": f.g1_from_lem(vkey[0:64]), "beta2": f.g2_from_lem(vkey[64:192]),
          "gamma2": f.g2_from_lem(vkey[192:320]), "delta2": f.g2_from_lem(vkey[320:448]),
          "IC": [f.g1_from_lem(vkey[448 + 64 * i:512 + 64 * i]) for i in range(514)]}
    prover = amd.Prover(out["zkey"])
    assert prover.info.n_public == 513
    proof, pub = prover.prove(out["wtns"])
    assert pub == [str(x) for x in example_public_signals()]
    assert amd.stringify(pub).startswith('[\n "0",\n "1",')     # 5f = 0101 1111, JSON.stringify(x, null, 1) layout
    pts = (f.g1_from_obj(proof["pi_a"]), f.g2_from_obj(proof["pi_b"]), f.g1_from_obj(proof["pi_c"]))
    assert g.verify(vk, [int(x) for x in pub], pts)
    prover.close()

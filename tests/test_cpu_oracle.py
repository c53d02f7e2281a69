"""CPU suite: the oracle pinned against everything there is to pin it against.
The reference holds no prove-boundary vectors (parity UNPINNED), so the pins are
(i) App. B constants re-derived here, (ii) algebraic self-consistency: prover == trapdoor KAT ==
pairing check, (iii) the committed golden fixtures, (iv) the reference's own witness-side golden
data for the MoH example pass (from /root/reference/test/nzcp.js:41-51)."""
import hashlib
import json
import random

import pytest

import bn254 as b
import formats as f
import groth16 as g
import synth
from conftest import golden_path


def test_constants_app_b():
    assert b.Q % 4 == 3 and (b.R - 1) % (1 << 28) == 0 and (b.R - 1) >> 28 & 1
    assert b.RR == 0x0e0a77c19a07df2f666ea36f7879462e36fc76959f60cd29ac96341c4ffffffb
    assert b.R2R == 0x0216d0b17f4e44a58c49833d53bb808553fe3ab1e35c59e31bb8e645ae216da7
    assert b.RQ == 0x0e0a77c19a07df2f666ea36f7879462c0a78eb28f5c70b3dd35d438dc58f0d9d
    assert b.R2Q == 0x06d89f71cab8351f47ab1eff0a417ff6b5e71911d44501fbf32cfc5b538afa89
    assert (-pow(b.R, -1, 1 << 32)) % (1 << 32) == 0xefffffff
    assert (-pow(b.Q, -1, 1 << 32)) % (1 << 32) == 0xe4866389
    assert b.FR_W28 == 19103219067921713944291392827692070036145651957329286315305642004821462161904
    assert pow(b.FR_W28, 1 << 27, b.R) == b.R - 1                       # order exactly 2^28
    assert all(pow(x, (b.R - 1) // 2, b.R) == 1 for x in (2, 3, 4)) and pow(5, (b.R - 1) // 2, b.R) == b.R - 1
    assert b.G1.on_curve(b.G1_GEN) and b.G2.on_curve(b.G2_GEN)
    assert b.G1.jmul(b.G1_GEN, b.R)[2] == 0 and b.G2.jmul(b.G2_GEN, b.R)[2] == (0, 0)
    # b' = 3/(9+u)
    assert b.f2_mul(b.G2_B, (9, 1)) == (3, 0)


def test_generated_header_matches_oracle_constants():
    import os, re
    hdr = open(os.path.join(os.path.dirname(golden_path("x")), "..", "..", "nzcp-circom_amd", "csrc",
                            "bn254_consts.h")).read()

    def limbs(name):
        m = re.search(r"#define %s\s+\{([^}]*)\}" % name, hdr)
        ws = [int(x.strip().rstrip("u"), 16) for x in m.group(1).split(",")]
        return sum(w << (32 * i) for i, w in enumerate(ws))
    assert limbs("G16_FQ_P") == b.Q and limbs("G16_FR_P") == b.R
    assert limbs("G16_FQ_ONE") == b.RQ and limbs("G16_FR_ONE") == b.RR
    assert limbs("G16_FQ_R2") == b.R2Q and limbs("G16_FR_R2") == b.R2R
    assert limbs("G16_FR_W28") == b.FR_W28 * b.RR % b.R
    assert limbs("G16_G2X0") == b.G2_GEN[0][0] * b.RQ % b.Q


def test_pairing_bilinear():
    rng = random.Random(3)
    k = rng.randrange(b.R)
    P, Qk = b.G1.mul(b.G1_GEN, k), b.G2.mul(b.G2_GEN, k)
    assert b.pairing_product_is_one([(P, b.G2_GEN), (b.G1.neg(b.G1_GEN), Qk)])
    assert not b.pairing_product_is_one([(P, b.G2_GEN), (b.G1_GEN, Qk)])


def test_ntt_kats():
    rng = random.Random(4)
    v = [rng.randrange(b.R) for _ in range(64)]
    assert g.ntt(g.ntt(v), inverse=True) == v
    w = b.fr_root(6)
    x = rng.randrange(b.R)   # Horner at a random power
    out = g.ntt(v)
    for i in (0, 5, 63):
        assert out[i] == sum(c * pow(w, i * j, b.R) for j, c in enumerate(v)) % b.R
    assert g.ntt([7] + [0] * 63) == [7] * 64


def test_coset_evaluation_semantics():
    """iNTT -> *w_2N^i -> NTT == evaluation at the odd 2N-th roots."""
    rng = random.Random(6)
    N = 16
    ev = [rng.randrange(b.R) for _ in range(N)]
    coef = g.ntt(ev, inverse=True)
    inc = b.fr_root(5)
    sh = g.ntt([c * pow(inc, i, b.R) % b.R for i, c in enumerate(coef)])
    for i in range(N):
        x = pow(inc, 2 * i + 1, b.R)
        assert sh[i] == sum(c * pow(x, j, b.R) for j, c in enumerate(coef)) % b.R


@pytest.mark.parametrize("n,p,m,seed", [(30, 2, 20, 5), (90, 7, 64, 6)])
def test_prover_equals_trapdoor_kat_and_verifies(n, p, m, seed):
    rows, w = synth.make(n, p, m, seed)
    assert synth.check_r1cs(rows, w)
    zk, sec = g.setup(n, p, rows, g.trapdoor(seed + 1))
    rng = synth.Xoshiro(seed + 2)
    r, s = rng.rand_fr(), rng.rand_fr()
    proof, pub = g.prove(zk, w, r, s)
    assert proof == g.expected_proof(sec, p, w, r, s)
    assert g.verify(zk, pub, proof)
    assert not g.verify(zk, [(pub[0] + 1) % b.R] + pub[1:], proof)
    # an unsatisfying witness yields a proof that does NOT verify (the KAT detects bad witnesses)
    w_bad = list(w)
    for sig in range(p + 1, n):
        w_bad = list(w); w_bad[sig] = (w_bad[sig] + 1) % b.R
        if not synth.check_r1cs(rows, w_bad):
            break
    proof_bad, pub_bad = g.prove(zk, w_bad, r, s)
    assert not g.verify(zk, pub_bad, proof_bad)


@pytest.mark.parametrize("name", ["tiny", "small", "nzcp513"])
def test_golden_fixtures_reproduce(name):
    zkb = open(golden_path(name + ".zkey"), "rb").read()
    wtb = open(golden_path(name + ".wtns"), "rb").read()
    meta = json.load(open(golden_path(name + ".json")))
    zk, w = f.read_zkey(zkb), f.read_wtns(wtb)["w"]
    assert (zk["nVars"], zk["nPublic"]) == (meta["n"], meta["p"])
    assert f.write_zkey(zk) == zkb and f.write_wtns(w) == wtb            # codecs round-trip
    proof, pub = g.prove(zk, w, int(meta["r"]), int(meta["s"]))
    assert f.proof_obj(*proof) == meta["proof"] and [str(x) for x in pub] == meta["public"]
    k = meta["kat"]
    assert proof == (b.G1.mul(b.G1_GEN, int(k["a"])), b.G2.mul(b.G2_GEN, int(k["b"])), b.G1.mul(b.G1_GEN, int(k["c"])))
    assert f.vkey_obj(zk) == meta["vkey"]
    if name != "nzcp513":
        assert g.verify(zk, pub, proof)


def test_nzcp513_public_layout_and_verify():
    """p = 513: w[1..256] | w[257..512] bits, w[513] = exp (/root/reference/test/nzcp.js:41-47)."""
    meta = json.load(open(golden_path("nzcp513.json")))
    pub = [int(x) for x in meta["public"]]
    assert len(pub) == 513 and all(x in (0, 1) for x in pub[:512]) and pub[512] == synth.EXP_EXAMPLE
    zk = f.read_zkey(open(golden_path("nzcp513.zkey"), "rb").read())
    pr = meta["proof"]
    assert g.verify(zk, pub, (f.g1_from_obj(pr["pi_a"]), f.g2_from_obj(pr["pi_b"]), f.g1_from_obj(pr["pi_c"])))


def test_example_pass_public_signals_app_d2():
    """The reference's own golden data for the MoH example pass: the sha256
    of ToBeSigned and of 'Jack,Sparrow,1960-04-16' -- what public.json must hold for the real circuit."""
    tbs = bytes.fromhex(
        "846a5369676e6174757265314aa204456b65792d3101264059011fa501781e6469643a7765623a6e7a63702e636f76696431392e6865616c74682e6e7a051a61819a0a041a7450400a627663a46840636f6e7465787482782668747470733a2f2f7777772e77332e6f72672f323031382f63726564656e7469616c732f7631782a68747470733a2f2f6e7a63702e636f76696431392e6865616c74682e6e7a2f636f6e74657874732f76316776657273696f6e65312e302e306474797065827456657269666961626c6543726564656e7469616c6f5075626c6963436f766964506173737163726564656e7469616c5375626a656374a369676976656e4e616d65644a61636b6a66616d696c794e616d656753706172726f7763646f626a313936302d30342d3136075060a4f54d4e304332be33ad78b1eafa4b")
    assert len(tbs) == 314
    assert hashlib.sha256(tbs).hexdigest() == "271ce33d671a2d3b816d788135f4343e14bc66802f8cd841faac939e8c11f3ee"
    assert hashlib.sha256(b"Jack,Sparrow,1960-04-16").hexdigest() == \
        "5fb355822221720ea4ce6734e5a09e459d452574a19310c0cea7c141f43a3dab"
    assert int.from_bytes(tbs[69:73], "big") == synth.EXP_EXAMPLE and tbs[68] == 0x1a


def test_binfile_errors_match_snarkjs():
    wt = open(golden_path("tiny.wtns"), "rb").read()
    with pytest.raises(ValueError, match="x: Invalid File format"):
        f.read_binfile(b"abcd" + wt[4:], "wtns", 2, "x")
    bad = bytearray(wt); bad[4] = 3
    with pytest.raises(ValueError, match="Version not supported"):
        f.read_binfile(bytes(bad), "wtns", 2)
    zk = bytearray(open(golden_path("tiny.zkey"), "rb").read())
    secs = f.read_binfile(bytes(zk), "zkey", 2)
    zk[secs[1][0][0]] = 2   # protocol id 2 = plonk
    with pytest.raises(ValueError, match="zkey file is not groth16"):
        f.read_zkey(bytes(zk))

"""The SHA-256 chain circuit (g16_sha256_chain_setup): a REAL constraint system.
Host-only checks: every R1CS row holds on the emitted witness (big-int arithmetic of the oracle), the public
outputs are the SHA-256 chain of the private message as computed by hashlib (an independent implementation),
the witness is bits only, and a flipped witness bit breaks a constraint."""
import hashlib

import bn254 as b
import formats as f


def _bits_msb_first(data):
    return [(byte >> (7 - k)) & 1 for byte in data for k in range(8)]


def _chain(msg, blocks):
    d = msg
    for _ in range(blocks):
        d = hashlib.sha256(d).digest()
    return d


def _check_rows(rows, w):
    for i, (A, B, C) in enumerate(rows):
        a = sum(cf * w[s] for s, cf in A) % b.R
        bb = sum(cf * w[s] for s, cf in B) % b.R
        c = sum(cf * w[s] for s, cf in C) % b.R
        if a * bb % b.R != c:
            return i
    return -1


def test_sha256_chain_r1cs_is_satisfied_and_outputs_match_hashlib(amd):
    msg = bytes(range(7, 39))
    for blocks in (1, 2):
        out = amd.sha256_chain_setup(blocks, msg, 1, want_zkey=False, want_r1cs=True)
        r1 = f.read_r1cs(out["r1cs"])
        w = f.read_wtns(out["wtns"])["w"]
        assert r1["prime"] == b.R and r1["nPubOut"] == 256 and r1["nPubIn"] == 0
        assert r1["nWires"] == len(w) and w[0] == 1
        assert set(w) <= {0, 1}
        assert 24000 * blocks < len(r1["rows"]) < 29000 * blocks + 600      # ~26-27 k constraints per compression
        assert w[1:257] == _bits_msb_first(_chain(msg, blocks))
        assert w[257:513] == _bits_msb_first(msg)
        assert _check_rows(r1["rows"], w) == -1
        # a wrong witness is caught: flip a private message bit / an internal gate output
        for wire in (300, len(w) // 2):
            bad = list(w)
            bad[wire] ^= 1
            assert _check_rows(r1["rows"], bad) >= 0


def test_sha256_chain_other_messages(amd):
    """All-zero / all-one / text messages, three compressions: outputs follow hashlib."""
    for msg in (bytes(32), b"\xff" * 32, b"abcdbcdecdefdefgefghfghighijhijk"):
        out = amd.sha256_chain_setup(3, msg, 1, want_zkey=False)
        w = f.read_wtns(out["wtns"])["w"]
        assert w[1:257] == _bits_msb_first(_chain(msg, 3))
        assert w[257:513] == _bits_msb_first(msg)

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


# ---- the ToBeSigned hash of the MoH example pass: the reference's own golden value (captured by
# running /root/reference/test/helpers/nzcp.js on the URI at /root/reference/test/nzcp.js:51)
EXAMPLE_TBS_SHA256 = "271ce33d671a2d3b816d788135f4343e14bc66802f8cd841faac939e8c11f3ee"


def example_to_be_signed():
    """URI -> COSE_Sign1 -> Sig_structure, restating /root/reference/test/helpers/nzcp.js:140-158: base32 body,
    CBOR tag 18, array(4) [protected bstr, {} , payload bstr, signature bstr]; ToBeSigned =
    ["Signature1", protected, h'', payload]."""
    import base64
    from conftest import golden_path
    uri = open(golden_path("example_pass_uri.txt")).read().strip()
    b32 = uri.split("/")[-1]
    raw = base64.b32decode(b32 + "=" * ((8 - len(b32) % 8) % 8))

    def bstr(buf, pos):
        ib = buf[pos]
        assert ib >> 5 == 2
        ai, pos = ib & 31, pos + 1
        if ai < 24:
            n = ai
        elif ai == 24:
            n, pos = buf[pos], pos + 1
        else:
            assert ai == 25
            n, pos = int.from_bytes(buf[pos:pos + 2], "big"), pos + 2
        return buf[pos:pos + n], pos + n

    def enc(bs):
        n = len(bs)
        return (bytes([0x40 | n]) if n < 24 else bytes([0x58, n]) if n < 256 else bytes([0x59]) + n.to_bytes(2, "big")) + bs

    assert raw[0] == 0xD2 and raw[1] == 0x84
    prot, pos = bstr(raw, 2)
    assert raw[pos] == 0xA0
    payload, _ = bstr(raw, pos + 1)
    return bytes([0x84, 0x6A]) + b"Signature1" + enc(prot) + bytes([0x40]) + enc(payload)


def test_sha256_message_circuit_reproduces_reference_golden_tobesigned_hash(amd):
    """Plain SHA-256 circuit over the example pass's 314-byte ToBeSigned (6 compressions): its public signals are
    the `toBeSignedSha256` slice [256..511] of the NZCP circuit's public.json (/root/reference/test/nzcp.js:41-47)
    and must equal the reference's golden value."""
    tbs = example_to_be_signed()
    assert len(tbs) == 314 and hashlib.sha256(tbs).hexdigest() == EXAMPLE_TBS_SHA256
    out = amd.sha256_message_setup(tbs, 1, want_zkey=False, want_r1cs=True)
    w = f.read_wtns(out["wtns"])["w"]
    r1 = f.read_r1cs(out["r1cs"])
    assert w[1:257] == _bits_msb_first(bytes.fromhex(EXAMPLE_TBS_SHA256))
    assert w[257:257 + 314 * 8] == _bits_msb_first(tbs)
    assert r1["nWires"] == len(w) and set(w) <= {0, 1}
    assert _check_rows(r1["rows"], w) == -1


def test_sha256_message_lengths_around_the_padding_boundary(amd):
    """55 / 56 / 64 bytes: the 0x80 byte and the length field move into a second block."""
    for n in (0, 1, 55, 56, 63, 64, 119, 120):
        msg = bytes((7 * i + n) & 0xFF for i in range(n))
        out = amd.sha256_message_setup(msg, 1, want_zkey=False)
        w = f.read_wtns(out["wtns"])["w"]
        assert w[1:257] == _bits_msb_first(hashlib.sha256(msg).digest()), n


# the three public values of the MoH example pass (/root/reference/test/nzcp.js:51 URI run through
# /root/reference/test/helpers/nzcp.js): public.json = 256 + 256 bits MSB-first, then exp
EXAMPLE_CREDSUBJ_SHA256 = "5fb355822221720ea4ce6734e5a09e459d452574a19310c0cea7c141f43a3dab"
EXAMPLE_EXP = 1951416330
EXAMPLE_SEGS = [(258, 4), (274, 7), (286, 10)]      # "Jack", "Sparrow", "1960-04-16" inside ToBeSigned
EXAMPLE_EXP_OFF = 69                                  # after the 0x1a at 68


def example_public_signals():
    return (_bits_msb_first(bytes.fromhex(EXAMPLE_CREDSUBJ_SHA256)) + _bits_msb_first(bytes.fromhex(EXAMPLE_TBS_SHA256))
            + [EXAMPLE_EXP])


def test_nzcp_fixed_layout_circuit_reproduces_the_reference_public_signals(amd):
    """All 513 public signals of the reference's example-pass test (/root/reference/test/nzcp.js:41-49) from a
    real R1CS over the pass's ToBeSigned: both SHA-256 digests and exp."""
    tbs = example_to_be_signed()
    for (o, n), want in zip(EXAMPLE_SEGS, (b"Jack", b"Sparrow", b"1960-04-16")):
        assert tbs[o:o + n] == want
    assert hashlib.sha256(b"Jack,Sparrow,1960-04-16").hexdigest() == EXAMPLE_CREDSUBJ_SHA256
    out = amd.nzcp_fixed_layout_setup(tbs, EXAMPLE_SEGS, EXAMPLE_EXP_OFF, 1, want_zkey=False, want_r1cs=True)
    w = f.read_wtns(out["wtns"])["w"]
    r1 = f.read_r1cs(out["r1cs"])
    assert r1["nPubOut"] == 513 and r1["nWires"] == len(w)
    assert w[1:514] == example_public_signals()
    assert w[514:514 + 314 * 8] == _bits_msb_first(tbs)
    assert _check_rows(r1["rows"], w) == -1
    # the outputs are bound to ToBeSigned: a different dob byte changes both digests, nothing else
    tbs2 = bytearray(tbs)
    tbs2[286] = ord("2")
    w2 = f.read_wtns(amd.nzcp_fixed_layout_setup(bytes(tbs2), EXAMPLE_SEGS, EXAMPLE_EXP_OFF, 1, want_zkey=False)["wtns"])["w"]
    assert w2[1:257] == _bits_msb_first(hashlib.sha256(b"Jack,Sparrow,2960-04-16").digest())
    assert w2[257:513] == _bits_msb_first(hashlib.sha256(bytes(tbs2)).digest()) and w2[513] == EXAMPLE_EXP
    # a witness that claims another exp breaks the linear row
    bad = list(w)
    bad[513] += 1
    assert _check_rows(r1["rows"], bad) >= 0

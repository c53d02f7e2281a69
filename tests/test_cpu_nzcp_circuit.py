"""The NZCP circuit library built natively (csrc/nzcp_gadgets.h), template by template, against the REFERENCE'S OWN
test vectors -- the only place the reference pins results for this part of the path:

    /root/reference/test/cbor.js          GetType / GetX (all 256 bytes) :15-35, GetV :38-108, DecodeUint23 :117-128,
                                          DecodeUint KATs :136-177, ReadType :185-217, SkipValueScalar :221-260,
                                          SkipValue scalars + arrays :262-366, ReadStringLength :368-382,
                                          StringEquals :384-406, ReadMapLength :408-431, CopyString :435-477
    /root/reference/test/quinSelector.js  QuinSelector(0..5) :23-66 and its out-of-range cases
    /root/reference/test/nzcp.js          FindVCAndExp 28 -> (76, 68) :89, FindCredSubj 77 -> 246 :144,
                                          ReadCredSubj at 247 :217, ConcatCredSubj :260-283, NZCPPubIdentity :305-316

Each case builds ONE template over the vector's inputs through the C ABI (g16_nzcp_gadget), which also evaluates
every emitted R1CS row on the computed witness; "isRejected" cases of the reference must come back as an
unsatisfied constraint.  The CBOR encoders below restate /root/reference/test/helpers/cbor.js:10-58 (data
generators of the vectors).  Finally the whole NZCPPubIdentity(0, 314, 0, 4, 2, 4, 5) system is built for the MoH
example pass and its 513 public wires must equal the reference's golden public signals."""
import hashlib

import pytest

import formats as f
from test_cpu_sha256_circuit import (EXAMPLE_CREDSUBJ_SHA256, EXAMPLE_EXP, EXAMPLE_TBS_SHA256, _bits_msb_first, _check_rows,
                                     example_public_signals, example_to_be_signed)


# ---- CBOR encoders of the vectors (helpers/cbor.js)
def enc_uint(v):
    if v <= 23:
        return [v]
    if v <= 0xFF:
        return [24, v]
    if v <= 0xFFFF:
        return [25, v >> 8, v & 255]
    assert v <= 0xFFFFFFFF
    return [26, v >> 24, (v >> 16) & 255, (v >> 8) & 255, v & 255]


def enc_int(v):
    x = enc_uint(v)
    return [(0 << 5) | x[0]] + x[1:]


def enc_str(s):
    x = enc_uint(len(s))
    return [(3 << 5) | x[0]] + x[1:] + [ord(c) for c in s]


def enc_arr(items):
    x = enc_uint(len(items))
    return [(4 << 5) | x[0]] + x[1:] + [b for it in items for b in it]


def enc_map(pairs):
    x = enc_uint(len(pairs))
    return [(5 << 5) | x[0]] + x[1:] + [b for k, v in pairs for b in (k + v)]


def pad(arr, n):
    return list(arr) + [0] * max(0, n - len(arr))


def rejected(amd, name, params, inputs):
    with pytest.raises(amd.G16Error, match="constraint not satisfied"):
        amd.nzcp_gadget(name, params, inputs)


# ---- cbor.js
def test_get_type_and_get_x_all_bytes(amd):
    for v in range(256):
        assert amd.nzcp_gadget("getType", [], [v])[0] == [v >> 5]
        assert amd.nzcp_gadget("getX", [], [v])[0] == [v & 31]
    rejected(amd, "getType", [], [256])          # "input MUST be a byte"


def test_get_v(amd):
    for n in (3, 4, 5):
        arr = list(range(1, n + 1))
        for pos in range(n):
            assert amd.nzcp_gadget("getV", [n], arr + [pos])[0] == [arr[pos]]
        rejected(amd, "getV", [n], arr + [n])    # QuinSelector: index < choices


def test_decode_uint23(amd):
    for v in range(256):
        x = v & 31
        if x <= 23:
            assert amd.nzcp_gadget("decodeUint23", [], [v])[0] == [x]
        else:
            rejected(amd, "decodeUint23", [], [v])


@pytest.mark.parametrize("bytes_,v,value,next_pos", [
    ([0, 0, 0, 0], 167, 7, 0), ([0, 0, 0, 0], 168, 8, 0),                   # x <= 23
    ([31, 0, 0, 0], 120, 31, 1), ([38, 0, 0, 0], 120, 38, 1),               # x == 24
    ([42, 69, 0, 0], 25, 10821, 2), ([69, 42, 0, 0], 25, 17706, 2),         # x == 25
    ([97, 218, 192, 48], 26, 1641726000, 4), ([98, 150, 3, 64], 26, 1653998400, 4),   # x == 26
])
def test_decode_uint_kats(amd, bytes_, v, value, next_pos):
    out, _ = amd.nzcp_gadget("decodeUint", [4], bytes_ + [0, v])
    assert out == [value, next_pos]


def test_read_type(amd):
    for pos in (2, 1, 0):
        for v in range(256):
            b = [0, 0, 0]
            b[pos] = v
            assert amd.nzcp_gadget("readType", [3], b + [pos])[0] == [pos + 1, v >> 5, v]


def test_skip_value_scalar_and_skip_value_scalars(amd):
    for name, params in (("skipValueScalar", [5]), ("skipValue", [5, 4])):
        for n in range(5):
            assert amd.nzcp_gadget(name, params, pad(enc_str("a" * n), 5) + [0])[0] == [n + 1]
        for value in list(range(24)) + [0xFF, 0xFFFF, 0xFFFFFFFF]:
            cbor = enc_int(value)
            assert amd.nzcp_gadget(name, params, pad(cbor, 5) + [0])[0] == [len(cbor)]


@pytest.mark.parametrize("items,maxlen", [
    ([enc_int(23)] * 3, 5), ([enc_int(23)] * 4, 5), ([enc_int(0xFF)] * 2, 5), ([enc_int(0xFFFF)], 5),
    ([enc_int(0xFFFFFFFF)], 6), ([enc_str("q")] * 2, 5), ([enc_str("qwe")], 5), ([enc_str("q"), enc_int(0xFF)], 5),
    ([enc_str("q"), enc_int(23), enc_int(23)], 5),
])
def test_skip_value_arrays(amd, items, maxlen):
    cbor = enc_arr(items)
    assert amd.nzcp_gadget("skipValue", [maxlen, 4], pad(cbor, maxlen) + [0])[0] == [len(cbor)]


def test_read_string_length_string_equals_read_map_length(amd):
    for n in range(5):
        out, _ = amd.nzcp_gadget("readStringLength", [5], pad(enc_str("a" * n), 5) + [0])
        assert out == [n, len(enc_int(n))]
    rejected(amd, "readStringLength", [5], pad(enc_int(3), 5) + [0])          # hardcore_assert(type, STRING)
    const = [ord(c) for c in "abcde"]
    assert amd.nzcp_gadget("stringEquals", [5, 5] + const, const + [0, 5])[0] == [1]
    for n in range(6):
        assert amd.nzcp_gadget("stringEquals", [5, 5] + const, pad([ord("b")] * n, 5) + [0, n])[0] == [0]
    for k in (1, 2, 3):
        pairs = [(enc_int(4), enc_int(5)), (enc_int(5), enc_int(4)), (enc_int(7), enc_int(3))][:k]
        assert amd.nzcp_gadget("readMapLength", [7], pad(enc_map(pairs), 7) + [0])[0][0] == k
    rejected(amd, "readMapLength", [7], pad(enc_str("ab"), 7) + [0])          # hardcore_assert(type, MAP)


def test_copy_string(amd):
    for s in ("", "ab", "abcd"):
        out, _ = amd.nzcp_gadget("copyString", [5, 4], pad(enc_str(s), 5) + [0])
        assert out == pad([ord(c) for c in s], 4) + [len(s) + 1, len(s)]


# ---- quinSelector.js
def test_quin_selector(amd):
    assert amd.nzcp_gadget("quinSelector", [0], [(1 << 64) - 1])[0] == [0]     # QuinSelector(0): out <== 0
    for n in range(1, 6):
        arr = list(range(1, n + 1))
        for idx in range(n):
            assert amd.nzcp_gadget("quinSelector", [n], arr + [idx])[0] == [arr[idx]]
        rejected(amd, "quinSelector", [n], arr + [n + 3])


# ---- nzcp.js on the MoH example pass
def test_find_vc_and_exp_example_pass(amd):
    tbs = example_to_be_signed()
    out, ncons = amd.nzcp_gadget("findVCAndExp", [314, 0, 4], list(tbs) + [28, 5])
    assert out == [76, 68]                                                   # /root/reference/test/nzcp.js:89
    assert list(tbs[68:73]) == enc_uint(EXAMPLE_EXP)                          # the expiry date sits there (:74-76)
    assert ncons > 10_000


def test_find_cred_subj_example_pass(amd):
    tbs = example_to_be_signed()
    assert amd.nzcp_gadget("findCredSubj", [314, 2, 4], list(tbs) + [77, 4])[0] == [246]   # nzcp.js:144


def test_read_cred_subj_example_pass(amd):
    tbs = example_to_be_signed()
    out, _ = amd.nzcp_gadget("readCredSubj", [314, 32], list(tbs) + [247, 3])               # nzcp.js:217
    want = []
    for s in ("Jack", "Sparrow", "1960-04-16"):
        want += pad([ord(c) for c in s], 32) + [len(s)]
    assert out == want
    rejected(amd, "readCredSubj", [314, 32], list(tbs) + [247, 4])           # hardcore_assert(mapLen, 3)


def test_concat_cred_subj(amd):
    m = 64
    for g, fa, d in (("Jack", "Sparrow", "1960-04-16"), ("A", "B", "2000-01-01"), ("Wilhelmina", "Featherstonehaugh", "1999-12-31")):
        inp = []
        for s in (g, fa, d):
            inp += pad([ord(c) for c in s], m) + [len(s)]
        out, _ = amd.nzcp_gadget("concatCredSubj", [m], inp)
        res = f"{g},{fa},{d}"
        assert out == pad([ord(c) for c in res], m) + [len(res)]


@pytest.mark.parametrize("n", [0, 1, 55, 56, 63, 64, 100, 119])
def test_sha256_var_matches_hashlib(amd, n):
    """The variable-length SHA-256 gadget (the contract of Sha256Var(1): 2 blocks) against hashlib, around the
    padding boundaries; a message that leaves no room for the padding is refused."""
    msg = bytes((11 * i + n) & 0xFF for i in range(n))
    out, _ = amd.nzcp_gadget("sha256Var", [1], [8 * n] + list(msg))
    assert out == _bits_msb_first(hashlib.sha256(msg).digest())


def test_sha256_var_rejects_overlong_and_unaligned(amd):
    msg = list(range(120))
    rejected(amd, "sha256Var", [1], [8 * 120] + msg)          # 120 + 1 + 8 bytes do not fit 2 blocks
    rejected(amd, "sha256Var", [1], [8 * 10 + 3] + msg[:11])  # not a whole number of bytes


def test_nzcp_pub_identity_example_pass_golden_public_signals(amd):
    """NZCPPubIdentity(0, 314, 0, 4, 2, 4, 5) -- /root/reference/circuits/nzcp_exampleTest.circom -- with the CBOR
    search IN the circuit: the 513 public wires of the natively built witness equal the reference's golden public
    signals of the MoH example pass (/root/reference/test/nzcp.js:41-49, :305-316), every R1CS row holds, and a
    tampered witness (another expiry, another name byte) breaks one."""
    tbs = example_to_be_signed()
    out = amd.nzcp_circuit_setup(amd.NZCP_EXAMPLE_PARAMS, tbs, 1, want_zkey=False, want_r1cs=True)
    w = f.read_wtns(out["wtns"])["w"]
    r1 = f.read_r1cs(out["r1cs"])
    assert r1["nPubOut"] == 513 and r1["nWires"] == len(w) and len(r1["rows"]) == out["n_constraints"]
    assert w[1:514] == example_public_signals()
    assert w[1:257] == _bits_msb_first(bytes.fromhex(EXAMPLE_CREDSUBJ_SHA256))
    assert w[257:513] == _bits_msb_first(bytes.fromhex(EXAMPLE_TBS_SHA256)) and w[513] == EXAMPLE_EXP
    assert w[514:514 + 314 * 8] == _bits_msb_first(tbs) and w[514 + 314 * 8] == 314
    assert _check_rows(r1["rows"], w) == -1
    bad = list(w)
    bad[513] += 1
    assert _check_rows(r1["rows"], bad) >= 0
    print(f"NZCPPubIdentity(example): {out['n_constraints']} constraints, {len(w)} wires")


def test_nzcp_pub_identity_other_passes_and_rejections(amd):
    """Same circuit, other ToBeSigned inputs: a pass whose credential subject differs (names found by the
    in-circuit search, not by fixed offsets: the strings move), and inputs the circuit must refuse."""
    tbs = bytearray(example_to_be_signed())
    # longer given name: re-encode the credentialSubject map in place (string lengths <= 23 keep one-byte heads)
    i = bytes(tbs).index(b"Jack")
    assert tbs[i - 1] == 0x60 + 4
    mod = bytes(tbs[:i - 1]) + bytes([0x60 + 6]) + b"Jackie" + bytes(tbs[i + 4:])
    # the enclosing byte-string head of the payload carries its length: fix it (0x59 hi lo)
    j = mod.index(b"\x59")
    ln = int.from_bytes(mod[j + 1:j + 3], "big") + 2
    mod = mod[:j + 1] + ln.to_bytes(2, "big") + mod[j + 3:]
    out = amd.nzcp_circuit_setup((0, 320, 0, 4, 2, 4, 5), mod, 1, want_zkey=False)
    w = f.read_wtns(out["wtns"])["w"]
    assert w[1:257] == _bits_msb_first(hashlib.sha256(b"Jackie,Sparrow,1960-04-16").digest())
    assert w[257:513] == _bits_msb_first(hashlib.sha256(mod).digest()) and w[513] == EXAMPLE_EXP
    # not a claims map at the expected offset -> ReadMapLength's hardcore_assert
    broken = bytearray(example_to_be_signed())
    broken[27] = 0x80
    with pytest.raises(amd.G16Error, match="constraint not satisfied"):
        amd.nzcp_circuit_setup(amd.NZCP_EXAMPLE_PARAMS, bytes(broken), 1, want_zkey=False)

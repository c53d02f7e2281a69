#!/usr/bin/env python3
"""Pass URI -> the fixed-layout NZCP interface circuit of that pass, ready for snarkjs cross-checks (host only):

    python tools/nzcp_fixed_circuit.py tests/golden/example_pass_uri.txt outdir [--seed N] [--prove]

writes outdir/circuit.r1cs, circuit.zkey (test-only KNOWN trapdoor), witness.wtns, verification_key.json and
expected_public.json (the 513 values /root/reference/test/nzcp.js:41-49 checks: SHA-256("given,family,dob") bits,
SHA-256(ToBeSigned) bits, exp).  With --prove (needs the MI355X) also proof.json / public.json, so that on a box
with snarkjs:   snarkjs groth16 verify outdir/verification_key.json outdir/public.json outdir/proof.json
and, for the drop-in claim itself,   snarkjs groth16 prove outdir/circuit.zkey outdir/witness.wtns p.json pub.json
must give the same public.json.  The circuit takes the string / exp offsets of THIS pass as constants
(include/g16_prover.h: g16_nzcp_fixed_layout_setup); the reference finds them by in-circuit CBOR parsing."""
import argparse
import base64
import hashlib
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as entry  # noqa: E402


def _bstr(buf, pos):
    ib = buf[pos]
    if ib >> 5 != 2:
        raise ValueError("not a CBOR byte string")
    ai, pos = ib & 31, pos + 1
    if ai < 24:
        n = ai
    elif ai == 24:
        n, pos = buf[pos], pos + 1
    elif ai == 25:
        n, pos = int.from_bytes(buf[pos:pos + 2], "big"), pos + 2
    else:
        raise ValueError("unsupported CBOR length")
    return buf[pos:pos + n], pos + n


def _enc_bstr(bs):
    n = len(bs)
    head = bytes([0x40 | n]) if n < 24 else bytes([0x58, n]) if n < 256 else bytes([0x59]) + n.to_bytes(2, "big")
    return head + bs


def to_be_signed(uri):
    """/root/reference/test/helpers/nzcp.js:140-158: base32 body -> COSE_Sign1 -> Sig_structure."""
    b32 = uri.strip().split("/")[-1]
    raw = base64.b32decode(b32 + "=" * ((8 - len(b32) % 8) % 8))
    if raw[0] != 0xD2 or raw[1] != 0x84:
        raise ValueError("not a COSE_Sign1 pass")
    prot, pos = _bstr(raw, 2)
    if raw[pos] != 0xA0:
        raise ValueError("unexpected unprotected header")
    payload, _ = _bstr(raw, pos + 1)
    return bytes([0x84, 0x6A]) + b"Signature1" + _enc_bstr(prot) + bytes([0x40]) + _enc_bstr(payload)


def _text_at(buf, pos):
    ib = buf[pos]
    if ib >> 5 != 3:
        raise ValueError("expected a CBOR text string")
    ai = ib & 31
    if ai < 24:
        return pos + 1, ai
    if ai == 24:
        return pos + 2, buf[pos + 1]
    raise ValueError("text string too long")


def fixed_layout(tbs):
    """offsets of givenName / familyName / dob values and of the 4 exp bytes (nzcpInput.js fixedLayout)."""
    subj = tbs.index(b"\x71credentialSubject")
    segs = []
    for key in (b"\x69givenName", b"\x6afamilyName", b"\x63dob"):
        k = tbs.index(key, subj)
        segs.append(_text_at(tbs, k + len(key)))
    e = tbs.index(b"\x04\x1a", 27)          # claim 4 (exp), 32-bit uint
    return segs, e + 2


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("uri_file")
    ap.add_argument("outdir")
    ap.add_argument("--seed", type=lambda x: int(x, 0), default=0x6E7A6370)
    ap.add_argument("--prove", action="store_true")
    a = ap.parse_args()
    amd = entry.load_package()
    tbs = to_be_signed(open(a.uri_file).read())
    segs, exp_off = fixed_layout(tbs)
    out = amd.nzcp_fixed_layout_setup(tbs, segs, exp_off, a.seed, want_r1cs=True)
    os.makedirs(a.outdir, exist_ok=True)
    for name, key in (("circuit.r1cs", "r1cs"), ("circuit.zkey", "zkey"), ("witness.wtns", "wtns")):
        with open(os.path.join(a.outdir, name), "wb") as fh:
            fh.write(out[key])
    vk = amd.vkey_json(out["vkey"], 513)
    with open(os.path.join(a.outdir, "verification_key.json"), "w") as fh:
        fh.write(amd.stringify(vk))
    subj = b",".join(tbs[o:o + n] for o, n in segs)
    bits = lambda d: [str((byte >> (7 - k)) & 1) for byte in d for k in range(8)]  # noqa: E731
    expected = bits(hashlib.sha256(subj).digest()) + bits(hashlib.sha256(tbs).digest()) + \
        [str(int.from_bytes(tbs[exp_off:exp_off + 4], "big"))]
    with open(os.path.join(a.outdir, "expected_public.json"), "w") as fh:
        fh.write(amd.stringify(expected))
    print(f"ToBeSigned {len(tbs)} bytes, credential subject {subj.decode()!r}, exp {expected[-1]}; "
          f"strings at {segs}, exp bytes at {exp_off}")
    if a.prove:
        prover = amd.Prover(out["zkey"])
        proof, pub = prover.prove(out["wtns"])
        prover.close()
        assert pub == expected, "public signals differ from the pass"
        with open(os.path.join(a.outdir, "proof.json"), "w") as fh:
            fh.write(amd.stringify(proof))
        with open(os.path.join(a.outdir, "public.json"), "w") as fh:
            fh.write(amd.stringify(pub))
        print("proved on the GPU: public.json == expected_public.json")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Synthetic NZ COVID Pass payloads in the LIVE format (real live passes are private: /root/reference/.env.example),
for tests and bench.py: the COSE Sig_structure `ToBeSigned` the NZCP circuit takes as input.

Restates the generator of nzcp-circom_amd/js/nzcpInput.js (syntheticPass) on the Python side.  Layout of a live pass
as the reference's tests name it (/root/reference/test/nzcp.js:103,158,230; nzcptpl.circom:438): protected header
with an 8-byte kid, claims map at byte 30 of ToBeSigned, 31-character issuer => exp at 72, vc at 80,
credentialSubject map at 250."""
import hashlib


def _head(major, n):
    if n < 24:
        return bytes([(major << 5) | n])
    if n < 256:
        return bytes([(major << 5) | 24, n])
    return bytes([(major << 5) | 25]) + n.to_bytes(2, "big")


def cbor_text(s):
    b = s.encode()
    return _head(3, len(b)) + b


def cbor_bytes(b):
    return _head(2, len(b)) + b


def cbor_u32(v):
    return bytes([0x1A]) + v.to_bytes(4, "big")


def to_be_signed(given="Jack", family="Sparrow", dob="1960-04-16", live=True, nbf=1635883530, exp=1951416330):
    iss = "did:web:nzcp.identity.health.nz" if live else "did:web:nzcp.covid19.health.nz"
    kid = b"z12Kf7UQ" if live else b"key-1"
    cti = hashlib.sha256(f"{given}|{family}|{dob}".encode()).digest()[:16]
    protected = bytes([0xA2, 0x04]) + cbor_bytes(kid) + bytes([0x01, 0x26])
    subj = bytes([0xA3]) + cbor_text("givenName") + cbor_text(given) + cbor_text("familyName") + cbor_text(family) + \
        cbor_text("dob") + cbor_text(dob)
    vc = bytes([0xA4]) + cbor_text("@context") + bytes([0x82]) + cbor_text("https://www.w3.org/2018/credentials/v1") + \
        cbor_text("https://nzcp.covid19.health.nz/contexts/v1") + cbor_text("version") + cbor_text("1.0.0") + \
        cbor_text("type") + bytes([0x82]) + cbor_text("VerifiableCredential") + cbor_text("PublicCovidPass") + \
        cbor_text("credentialSubject") + subj
    payload = bytes([0xA5, 0x01]) + cbor_text(iss) + bytes([0x05]) + cbor_u32(nbf) + bytes([0x04]) + cbor_u32(exp) + \
        cbor_text("vc") + vc + bytes([0x07]) + cbor_bytes(cti)
    return bytes([0x84]) + cbor_text("Signature1") + cbor_bytes(protected) + cbor_bytes(b"") + cbor_bytes(payload)


if __name__ == "__main__":
    t = to_be_signed()
    print(len(t), t.hex())

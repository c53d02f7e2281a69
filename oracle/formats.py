"""ORACLE (test infrastructure) -- iden3 binfile / .wtns / .zkey / proof.json codecs.

PARITY UNPINNED: no .zkey/.wtns/proof.json fixture exists in /root/reference
(/root/reference/.gitignore:2-4,15-16 excludes them).  Layouts restate the published formats of
@iden3/binfileutils 0.0.10 (yarn.lock:10-16) and snarkjs 0.4.12 zkey_utils.js / wtns_utils.js
(yarn.lock:987-1001) as recorded in SURVEY.md App. A.

All integers little-endian; field elements are 32 bytes.
"""
import json
import struct

from bn254 import Q, R, RQ, RR, R2R

N8 = 32


def le(x, n=N8):
    return int(x).to_bytes(n, "little")


def from_le(b):
    return int.from_bytes(b, "little")


# ------------------------------------------------------------------ binfile container (App. A.1)
def write_binfile(magic, version, sections):
    """sections: list of (id, bytes) in the order they should appear."""
    out = [magic.encode("ascii"), struct.pack("<II", version, len(sections))]
    for sid, payload in sections:
        out.append(struct.pack("<IQ", sid, len(payload)))
        out.append(payload)
    return b"".join(out)


def read_binfile(buf, magic, max_version, name="file"):
    """Mirrors readBinFile: errors '<name>: Invalid File format' / 'Version not supported'."""
    if len(buf) < 12 or buf[:4] != magic.encode("ascii"):
        raise ValueError(f"{name}: Invalid File format")
    version, nsec = struct.unpack_from("<II", buf, 4)
    if version > max_version:
        raise ValueError("Version not supported")
    pos = 12
    sections = {}
    for _ in range(nsec):
        sid, size = struct.unpack_from("<IQ", buf, pos)
        pos += 12
        sections.setdefault(sid, []).append((pos, size))
        pos += size
    return sections


def section(buf, sections, sid):
    pos, size = sections[sid][0]
    return buf[pos:pos + size]


# ------------------------------------------------------------------ points (LEM = LE Montgomery)
def g1_to_lem(P):
    if P is None:
        return bytes(64)
    return le(P[0] * RQ % Q) + le(P[1] * RQ % Q)


def g2_to_lem(P):
    if P is None:
        return bytes(128)
    (x0, x1), (y0, y1) = P
    return b"".join(le(v * RQ % Q) for v in (x0, x1, y0, y1))


_RQ_INV = pow(RQ, -1, Q)


def g1_from_lem(b):
    if b == bytes(64):
        return None
    return (from_le(b[:32]) * _RQ_INV % Q, from_le(b[32:64]) * _RQ_INV % Q)


def g2_from_lem(b):
    if b == bytes(128):
        return None
    v = [from_le(b[i * 32:(i + 1) * 32]) * _RQ_INV % Q for i in range(4)]
    return ((v[0], v[1]), (v[2], v[3]))


# ------------------------------------------------------------------ .r1cs (App. A.4)
def write_r1cs(n_wires, n_pub_out, n_pub_in, rows):
    """rows: [(A, B, C)] with terms [(wire, coef)]; iden3 r1cs v1 (header, constraints, wire map)."""
    n_prv = 0
    hdr = (struct.pack("<I", N8) + le(R) + struct.pack("<IIII", n_wires, n_pub_out, n_pub_in, n_prv) +
           struct.pack("<Q", n_wires) + struct.pack("<I", len(rows)))
    body = []
    for row in rows:
        for lc in row:
            body.append(struct.pack("<I", len(lc)))
            for wire, cf in lc:
                body.append(struct.pack("<I", wire) + le(cf % R))
    wmap = b"".join(struct.pack("<Q", i) for i in range(n_wires))
    return write_binfile("r1cs", 1, [(1, hdr), (2, b"".join(body)), (3, wmap)])


def read_r1cs(buf, name="r1cs"):
    """-> dict(nWires, nPubOut, nPubIn, nPrvIn, rows=[(A, B, C)] with terms [(wire, coef)])."""
    secs = read_binfile(buf, "r1cs", 1, name)
    h = section(buf, secs, 1)
    n8 = struct.unpack_from("<I", h, 0)[0]
    prime = from_le(h[4:4 + n8])
    nw, npo, npi, nprv = struct.unpack_from("<IIII", h, 4 + n8)
    ncons = struct.unpack_from("<I", h, 4 + n8 + 16 + 8)[0]
    body = section(buf, secs, 2)
    pos = 0
    rows = []
    for _ in range(ncons):
        row = []
        for _k in range(3):
            nt = struct.unpack_from("<I", body, pos)[0]
            pos += 4
            lc = []
            for _t in range(nt):
                wire = struct.unpack_from("<I", body, pos)[0]
                lc.append((wire, from_le(body[pos + 4:pos + 4 + n8])))
                pos += 4 + n8
            row.append(lc)
        rows.append(tuple(row))
    return {"n8": n8, "prime": prime, "nWires": nw, "nPubOut": npo, "nPubIn": npi, "nPrvIn": nprv, "rows": rows}


# ------------------------------------------------------------------ .wtns (App. A.2)
def write_wtns(witness):
    hdr = struct.pack("<I", N8) + le(R) + struct.pack("<I", len(witness))
    body = b"".join(le(w) for w in witness)
    return write_binfile("wtns", 2, [(1, hdr), (2, body)])


def read_wtns(buf, name="wtns"):
    secs = read_binfile(buf, "wtns", 2, name)
    h = section(buf, secs, 1)
    n8 = struct.unpack_from("<I", h, 0)[0]
    q = from_le(h[4:4 + n8])
    nw = struct.unpack_from("<I", h, 4 + n8)[0]
    body = section(buf, secs, 2)
    return {"n8": n8, "q": q, "nWitness": nw,
            "w": [from_le(body[i * n8:(i + 1) * n8]) for i in range(nw)]}


# ------------------------------------------------------------------ .zkey (App. A.3)
def write_zkey(zk):
    """zk: dict with nVars, nPublic, domainSize, alpha1, beta1, beta2, gamma2, delta1, delta2,
    IC[], coefs[(m,c,s,value)], A[], B1[], B2[], C[], H[] -- points affine/None, values in Fr."""
    s1 = struct.pack("<I", 1)
    s2 = (struct.pack("<I", N8) + le(Q) + struct.pack("<I", N8) + le(R) +
          struct.pack("<III", zk["nVars"], zk["nPublic"], zk["domainSize"]) +
          g1_to_lem(zk["alpha1"]) + g1_to_lem(zk["beta1"]) + g2_to_lem(zk["beta2"]) +
          g2_to_lem(zk["gamma2"]) + g1_to_lem(zk["delta1"]) + g2_to_lem(zk["delta2"]))
    s3 = b"".join(g1_to_lem(P) for P in zk["IC"])
    s4 = struct.pack("<I", len(zk["coefs"])) + b"".join(
        struct.pack("<III", m, c, s) + le(v * R2R % R) for (m, c, s, v) in zk["coefs"])
    s5 = b"".join(g1_to_lem(P) for P in zk["A"])
    s6 = b"".join(g1_to_lem(P) for P in zk["B1"])
    s7 = b"".join(g2_to_lem(P) for P in zk["B2"])
    s8 = b"".join(g1_to_lem(P) for P in zk["C"])
    s9 = b"".join(g1_to_lem(P) for P in zk["H"])
    s10 = bytes(64) + struct.pack("<I", 0)
    return write_binfile("zkey", 1, [(1, s1), (2, s2), (3, s3), (4, s4), (5, s5), (6, s6),
                                     (7, s7), (8, s8), (9, s9), (10, s10)])


def read_zkey(buf, name="zkey"):
    secs = read_binfile(buf, "zkey", 2, name)
    proto = struct.unpack_from("<I", section(buf, secs, 1), 0)[0]
    if proto != 1:
        raise ValueError("zkey file is not groth16")
    h = section(buf, secs, 2)
    pos = 0
    n8q = struct.unpack_from("<I", h, pos)[0]; pos += 4
    q = from_le(h[pos:pos + n8q]); pos += n8q
    n8r = struct.unpack_from("<I", h, pos)[0]; pos += 4
    r = from_le(h[pos:pos + n8r]); pos += n8r
    nVars, nPublic, domainSize = struct.unpack_from("<III", h, pos); pos += 12
    zk = {"n8q": n8q, "q": q, "n8r": n8r, "r": r, "nVars": nVars, "nPublic": nPublic,
          "domainSize": domainSize}
    zk["alpha1"] = g1_from_lem(h[pos:pos + 64]); pos += 64
    zk["beta1"] = g1_from_lem(h[pos:pos + 64]); pos += 64
    zk["beta2"] = g2_from_lem(h[pos:pos + 128]); pos += 128
    zk["gamma2"] = g2_from_lem(h[pos:pos + 128]); pos += 128
    zk["delta1"] = g1_from_lem(h[pos:pos + 64]); pos += 64
    zk["delta2"] = g2_from_lem(h[pos:pos + 128]); pos += 128

    def g1s(sid):
        b = section(buf, secs, sid)
        return [g1_from_lem(b[i:i + 64]) for i in range(0, len(b), 64)]

    zk["IC"] = g1s(3)
    c = section(buf, secs, 4)
    nc = struct.unpack_from("<I", c, 0)[0]
    r2inv = pow(R2R, -1, R)
    coefs = []
    for i in range(nc):
        m, cc, s = struct.unpack_from("<III", c, 4 + i * 44)
        raw = from_le(c[4 + i * 44 + 12:4 + i * 44 + 44])
        coefs.append((m, cc, s, raw * r2inv % R))
    zk["coefs"] = coefs
    zk["A"], zk["B1"], zk["C"], zk["H"] = g1s(5), g1s(6), g1s(8), g1s(9)
    b7 = section(buf, secs, 7)
    zk["B2"] = [g2_from_lem(b7[i:i + 128]) for i in range(0, len(b7), 128)]
    return zk


# ------------------------------------------------------------------ JSON (App. A.5)
def _g1_obj(P):
    return ["0", "1", "0"] if P is None else [str(P[0]), str(P[1]), "1"]


def _g2_obj(P):
    if P is None:
        return [["0", "0"], ["1", "0"], ["0", "0"]]
    return [[str(P[0][0]), str(P[0][1])], [str(P[1][0]), str(P[1][1])], ["1", "0"]]


def proof_obj(A, B, C):
    return {"pi_a": _g1_obj(A), "pi_b": _g2_obj(B), "pi_c": _g1_obj(C),
            "protocol": "groth16", "curve": "bn128"}


def js_stringify(obj):
    """Byte-for-byte JSON.stringify(obj, null, 1)."""
    return json.dumps(obj, indent=1, separators=(",", ": "))


def vkey_obj(zk):
    return {"protocol": "groth16", "curve": "bn128", "nPublic": zk["nPublic"],
            "vk_alpha_1": _g1_obj(zk["alpha1"]), "vk_beta_2": _g2_obj(zk["beta2"]),
            "vk_gamma_2": _g2_obj(zk["gamma2"]), "vk_delta_2": _g2_obj(zk["delta2"]),
            "IC": [_g1_obj(P) for P in zk["IC"]]}


def g1_from_obj(o):
    return None if o[2] == "0" else (int(o[0]), int(o[1]))


def g2_from_obj(o):
    if o[2][0] == "0" and o[2][1] == "0":
        return None
    return ((int(o[0][0]), int(o[0][1])), (int(o[1][0]), int(o[1][1])))

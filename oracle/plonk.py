"""ORACLE (test infrastructure) -- PLONK setup / prove / verify on Python ints, in the snarkjs 0.4.12 convention.

PARITY UNPINNED against the reference: /root/reference scripts `snarkjs plonk setup / prove` (Makefile:30-33) but
holds no PLONK zkey, proof or verification key, and snarkjs itself is absent from this image.  This file
restates the published algorithm of snarkjs 0.4.12 ([EXT], pin /root/reference/yarn.lock:987-1001):

  plonk_setup.js   R1CS -> PLONK gates (public-input gates first; every linear combination reduced to one signal by
                   addition gates; q_M a b + q_L a + q_R b + q_O c + q_C + PI = 0), copy-constraint permutation over
                   the cosets {1, k1, k2} H, selector / sigma / Lagrange polynomials as N coefficients + 4N evaluations,
                   the first N + 6 powers of tau
  plonk_prove.js   rounds 1-5 with the nine blinding scalars b1..b9, Keccak-256 transcript (hashToFr of uncompressed
                   big-endian points / big-endian field elements), quotient computed on the 4N domain with the
                   blinding terms tracked separately (the "Tz" polynomial), T split into T1, T2, T3, linearisation
                   polynomial r with its evaluation IN the proof (eval_r), openings W_xi and W_xiw
  plonk_verify.js  the same challenges, t(xi) from the quotient identity, one pairing equation

Independent pins: the verifier below is derived from the KZG opening identity, not from the prover's code path -- a
proof only verifies if every round of the prover is consistent with it; Keccak-256 is checked against known digests;
the product's GPU prover (csrc/plonk.hip) must reproduce this prover's proof bit for bit under the same blinding.
"""
from bn254 import R, G1, G1_GEN, G2_GEN, G2, fr_root, pairing_product_is_one
from groth16 import ntt, log2_exact

# ------------------------------------------------------------------ Keccak-256 (original padding 0x01, not SHA-3's 0x06)
_RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B,
       0x0000000080000001, 0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088,
       0x0000000080008009, 0x000000008000000A, 0x000000008000808B, 0x800000000000008B, 0x8000000000008089,
       0x8000000000008003, 0x8000000000008002, 0x8000000000000080, 0x000000000000800A, 0x800000008000000A,
       0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
_ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]
_M64 = (1 << 64) - 1


def _rol(x, n):
    n %= 64
    return ((x << n) | (x >> (64 - n))) & _M64 if n else x


def _keccak_f(a):
    for rc in _RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], _ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(data):
    rate = 136
    msg = bytearray(data) + b"\x01"
    while len(msg) % rate:
        msg.append(0)
    msg[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(msg), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(msg[off + 8 * i:off + 8 * i + 8], "little")
        a = _keccak_f(a)
    out = b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))
    return out


def hash_to_fr(transcript):
    return int.from_bytes(keccak256(transcript), "big") % R


def g1_be(P):
    """G1.toRprUncompressed: x | y big-endian standard form; infinity = zeros."""
    return bytes(64) if P is None else P[0].to_bytes(32, "big") + P[1].to_bytes(32, "big")


def fr_be(x):
    return (x % R).to_bytes(32, "big")


# ------------------------------------------------------------------ R1CS -> PLONK gates (plonk_setup.js processConstraints)
def r1cs_to_plonk(n_vars, n_public, rows):
    """rows: [(A, B, C)] of [(wire, coef)] with <A,w><B,w> = <C,w>.  -> (gates, additions, plonk_n_vars);
    gate = (sl, sr, so, qm, ql, qr, qo, qc); addition = (s1, s2, f1, f2): w[new] = f1 w[s1] + f2 w[s2]."""
    gates, adds = [], []
    nv = [n_vars]
    for s in range(1, n_public + 1):
        gates.append((s, 0, 0, 0, 1, 0, 0, 0))

    def lc_of(terms):
        d = {}
        for wire, cf in terms:
            d[wire] = (d.get(wire, 0) + cf) % R
        # (snarkjs keeps a linear combination as a JS object keyed by signal id: iteration is in ascending id order)
        return {s: d[s] for s in sorted(d) if d[s]}

    def reduce_coefs(lc, max_c):
        k = lc.get(0, 0)
        cs = [(s, c) for s, c in lc.items() if s != 0 and c]
        while len(cs) > max_c:
            (s1, c1), (s2, c2) = cs[0], cs[1]
            cs = cs[2:]
            so = nv[0]
            nv[0] += 1
            gates.append((s1, s2, so, 0, (-c1) % R, (-c2) % R, 1, 0))
            adds.append((s1, s2, c1, c2))
            cs.append((so, 1))
        ss = [s for s, _ in cs] + [0] * (max_c - len(cs))
        cf = [c for _, c in cs] + [0] * (max_c - len(cs))
        return k, ss, cf

    def add_sum(lc):
        k, ss, cf = reduce_coefs(lc, 3)
        gates.append((ss[0], ss[1], ss[2], 0, cf[0], cf[1], cf[2], k))

    def lc_type(lc):
        n = sum(1 for s in lc if s != 0)
        return "n" if n else ("k" if lc.get(0, 0) else "0")

    for A, B, C in rows:
        a, b, c = lc_of(A), lc_of(B), lc_of(C)
        ta, tb = lc_type(a), lc_type(b)
        if ta == "0" or tb == "0":
            add_sum(c)
        elif ta == "k" or tb == "k":
            kk, other = (a[0], b) if ta == "k" else (b[0], a)
            j = dict(c)                       # kk * other - C = 0
            j = {s: (-v) % R for s, v in j.items()}
            for s, v in other.items():
                j[s] = (j.get(s, 0) + kk * v) % R
            add_sum({s: j[s] for s in sorted(j) if j[s]})
        else:
            ka, sa, ca = reduce_coefs(a, 1)
            kb, sb, cb = reduce_coefs(b, 1)
            kc, sc, cc = reduce_coefs(c, 1)
            gates.append((sa[0], sb[0], sc[0], ca[0] * cb[0] % R, ca[0] * kb % R, ka * cb[0] % R, (-cc[0]) % R,
                          (ka * kb - kc) % R))
    return gates, adds, nv[0]


def extend_witness(w, adds):
    w = list(w)
    for s1, s2, f1, f2 in adds:
        w.append((f1 * w[s1] + f2 * w[s2]) % R)
    return w


def check_gates(gates, n_public, w):
    """every gate holds on the extended witness (public-input gates: q_L a - pub = 0)"""
    for i, (sl, sr, so, qm, ql, qr, qo, qc) in enumerate(gates):
        a, b, c = w[sl], w[sr], w[so]
        pi = -w[i + 1] if i < n_public else 0
        if (qm * a * b + ql * a + qr * b + qo * c + qc + pi) % R:
            return False
    return True


# ------------------------------------------------------------------ setup
def _pad4(coefs, n):
    return ntt(list(coefs) + [0] * (3 * n))


def setup(n_vars, n_public, rows, tau):
    gates, adds, pnv = r1cs_to_plonk(n_vars, n_public, rows)
    power = max(3, (len(gates) - 1).bit_length()) if len(gates) > 1 else 3   # t has 3n + 6 coefficients: they must fit 4n
    while (1 << power) < len(gates):
        power += 1
    n = 1 << power
    w1 = fr_root(power)
    # k1, k2: smallest values whose cosets k H are disjoint from H and from each other (x in H <=> x^n = 1)
    def included(k, others):
        if pow(k, n, R) == 1:
            return True
        return any(pow(k * pow(o, -1, R) % R, n, R) == 1 for o in others)
    k1 = 2
    while included(k1, []):
        k1 += 1
    k2 = k1 + 1
    while included(k2, [k1]):
        k2 += 1
    cols = [[g[c] for g in gates] + [0] * (n - len(gates)) for c in range(3)]
    q = [[g[3 + c] for g in gates] + [0] * (n - len(gates)) for c in range(5)]
    # sigma: every appearance of a signal points at the previous one (value k_col w^row), the first at the last
    sigma = [0] * (3 * n)
    last, first = {}, {}
    w = 1
    for i in range(n):
        for col in range(3):
            s = cols[col][i]
            p = col * n + i
            if s in last:
                sigma[p] = last[s]
            else:
                first[s] = p
            last[s] = (w, w * k1 % R, w * k2 % R)[col]
        w = w * w1 % R
    for s, p in first.items():
        sigma[p] = last[s]
    srs = [G1.mul(G1_GEN, pow(tau, i, R)) for i in range(n + 6)]

    def commit(coefs):
        return G1.msm(srs[:len(coefs)], coefs)

    def pol(evals):
        c = ntt(evals, inverse=True)
        return c, _pad4(c, n)
    zk = {"protocol": "plonk", "nVars": pnv, "nPublic": n_public, "domainSize": n, "power": power, "nAdditions": len(adds),
          "nConstraints": len(gates), "k1": k1, "k2": k2, "additions": adds, "maps": cols, "srs": srs,
          "X_2": G2.mul(G2_GEN, tau)}
    for name, col in zip(("Qm", "Ql", "Qr", "Qo", "Qc"), q):
        c, e4 = pol(col)
        zk["pol_" + name], zk["ext_" + name], zk[name] = c, e4, commit(c)
    for k in range(3):
        c, e4 = pol(sigma[k * n:(k + 1) * n])
        zk["pol_S%d" % (k + 1)], zk["ext_S%d" % (k + 1)], zk["S%d" % (k + 1)] = c, e4, commit(c)
    zk["lagrange"] = []
    for j in range(max(n_public, 1)):
        c, e4 = pol([1 if i == j else 0 for i in range(n)])
        zk["lagrange"].append((c, e4))
    return zk


def vkey(zk):
    return {k: zk[k] for k in ("protocol", "nPublic", "power", "k1", "k2", "Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3", "X_2")}


# ------------------------------------------------------------------ prover (plonk_prove.js)
def _eval_pol(coefs, x):
    r = 0
    for c in reversed(coefs):
        r = (r * x + c) % R
    return r


def _div_pol1(P, d):
    n = len(P)
    res = [0] * n
    res[n - 2] = P[n - 1]
    for i in range(n - 3, -1, -1):
        res[i] = (P[i + 1] + d * res[i + 1]) % R
    assert P[0] % R == (-d * res[0]) % R, "Polinomial does not divide"
    return res


def prove(zk, witness, b):
    """witness: the circuit's wires w[0..nVars-nAdditions) (w[0] = 1); b: dict 1..9 of blinding scalars.
    -> (proof dict of ints / affine points, public signals)."""
    n, power = zk["domainSize"], zk["power"]
    k1, k2 = zk["k1"], zk["k2"]
    w1, w4 = fr_root(power), fr_root(power + 2)
    srs = zk["srs"]
    wit = list(witness)
    wit[0] = 0                      # "first element in plonk is not used": set to zero
    wit = extend_witness(wit, zk["additions"])
    A, B, C = ([wit[s] for s in zk["maps"][c]] for c in range(3))

    def exp_tau(coefs):
        return G1.msm(srs[:len(coefs)], [c % R for c in coefs])

    def to4t(ev, pz):
        a = ntt(ev, inverse=True)
        a4 = ntt(a + [0] * (3 * n))
        a1 = a + [0] * len(pz)
        for i, z in enumerate(pz):
            a1[n + i] = (a1[n + i] + z) % R
            a1[i] = (a1[i] - z) % R
        return a1, a4
    proof = {}
    # round 1
    pol_a, A4 = to4t(A, [b[2], b[1]])
    pol_b, B4 = to4t(B, [b[4], b[3]])
    pol_c, C4 = to4t(C, [b[6], b[5]])
    proof["A"], proof["B"], proof["C"] = exp_tau(pol_a), exp_tau(pol_b), exp_tau(pol_c)
    # round 2
    beta = hash_to_fr(g1_be(proof["A"]) + g1_be(proof["B"]) + g1_be(proof["C"]))
    gamma = hash_to_fr(fr_be(beta))
    S1e, S2e, S3e = zk["ext_S1"], zk["ext_S2"], zk["ext_S3"]
    num, den = [1] * n, [1] * n
    w = 1
    for i in range(n):
        nn = (A[i] + beta * w + gamma) * (B[i] + k1 * beta * w + gamma) * (C[i] + k2 * beta * w + gamma) % R
        dd = (A[i] + beta * S1e[4 * i] + gamma) * (B[i] + beta * S2e[4 * i] + gamma) * (C[i] + beta * S3e[4 * i] + gamma) % R
        num[(i + 1) % n] = num[i] * nn % R
        den[(i + 1) % n] = den[i] * dd % R
        w = w * w1 % R
    Z = [x * pow(y, -1, R) % R for x, y in zip(num, den)]
    assert Z[0] == 1, "Copy constraints does not match"
    pol_z, Z4 = to4t(Z, [b[9], b[8], b[7]])
    proof["Z"] = exp_tau(pol_z)
    # round 3
    alpha = hash_to_fr(g1_be(proof["Z"]))
    i4 = fr_root(2)                                   # Fr.w[2]: the primitive 4th root
    Z1 = [0, (-1 + i4) % R, (-2) % R, (-1 - i4) % R]
    Z2 = [0, (-2 * i4) % R, 4, (2 * i4) % R]
    Z3 = [0, (2 + 2 * i4) % R, (-8) % R, (2 - 2 * i4) % R]
    Qe = [zk["ext_" + k] for k in ("Qm", "Ql", "Qr", "Qo", "Qc")]
    L = zk["lagrange"]
    T, Tz = [0] * (4 * n), [0] * (4 * n)
    w = 1
    for i in range(4 * n):
        a, bb, c, z, zw = A4[i], B4[i], C4[i], Z4[i], Z4[(i + 4) % (4 * n)]
        qm, ql, qr, qo, qc = (Qe[k][i] for k in range(5))
        s1, s2, s3 = S1e[i], S2e[i], S3e[i]
        ap, bp, cp = (b[2] + b[1] * w) % R, (b[4] + b[3] * w) % R, (b[6] + b[5] * w) % R
        zp = (b[7] * w * w + b[8] * w + b[9]) % R
        ww = w * w1 % R
        zwp = (b[7] * ww * ww + b[8] * ww + b[9]) % R
        pl = 0
        for j in range(zk["nPublic"]):
            pl = (pl - L[j][1][i] * A[j]) % R
        p = i % 4

        def mul2(x, y, xp, yp):
            r = x * y % R
            rz = (x * yp + xp * y) % R
            if p:
                rz = (rz + Z1[p] * xp * yp) % R
            return r, rz

        def mul4(x, y, u, v, xp, yp, up, vp):
            r = x * y * u * v % R
            a0 = (xp * y * u * v + x * yp * u * v + x * y * up * v + x * y * u * vp) % R
            a1 = (xp * yp * u * v + xp * y * up * v + xp * y * u * vp + x * yp * up * v + x * yp * u * vp + x * y * up * vp) % R
            a2 = (x * yp * up * vp + xp * y * up * vp + xp * yp * u * vp + xp * yp * up * v) % R
            a3 = xp * yp * up * vp % R
            rz = a0
            if p:
                rz = (rz + Z1[p] * a1 + Z2[p] * a2 + Z3[p] * a3) % R
            return r, rz
        e1, e1z = mul2(a, bb, ap, bp)
        e1, e1z = e1 * qm % R, e1z * qm % R
        e1 = (e1 + a * ql + bb * qr + c * qo + pl + qc) % R
        e1z = (e1z + ap * ql + bp * qr + cp * qo) % R
        bw = beta * w % R
        e2, e2z = mul4((a + bw + gamma) % R, (bb + bw * k1 + gamma) % R, (c + bw * k2 + gamma) % R, z, ap, bp, cp, zp)
        e3, e3z = mul4((a + beta * s1 + gamma) % R, (bb + beta * s2 + gamma) % R, (c + beta * s3 + gamma) % R, zw,
                       ap, bp, cp, zwp)
        l1 = L[0][1][i]
        e4 = (z - 1) * l1 % R * alpha % R * alpha % R
        e4z = zp * l1 % R * alpha % R * alpha % R
        T[i] = (e1 + alpha * e2 - alpha * e3 + e4) % R
        Tz[i] = (e1z + alpha * e2z - alpha * e3z + e4z) % R
        w = w * w4 % R
    t = ntt(T, inverse=True)
    for i in range(n):
        t[i] = (-t[i]) % R
    for i in range(n, 4 * n):
        t[i] = (t[i - n] - t[i]) % R
        if i > 3 * n - 4:
            assert t[i] == 0, "T Polynomial is not divisible"
    tz = ntt(Tz, inverse=True)
    for i in range(4 * n):
        if i > 3 * n + 5:
            assert tz[i] == 0, "Tz Polynomial is not well calculated"
        else:
            t[i] = (t[i] + tz[i]) % R
    pol_t = t[:3 * n + 6]
    proof["T1"], proof["T2"], proof["T3"] = exp_tau(t[:n]), exp_tau(t[n:2 * n]), exp_tau(t[2 * n:3 * n + 6])
    # round 4
    xi = hash_to_fr(g1_be(proof["T1"]) + g1_be(proof["T2"]) + g1_be(proof["T3"]))
    ev = {"a": _eval_pol(pol_a, xi), "b": _eval_pol(pol_b, xi), "c": _eval_pol(pol_c, xi),
          "s1": _eval_pol(zk["pol_S1"], xi), "s2": _eval_pol(zk["pol_S2"], xi), "t": _eval_pol(pol_t, xi),
          "zw": _eval_pol(pol_z, xi * w1 % R)}
    coef_ab = ev["a"] * ev["b"] % R
    e2 = (ev["a"] + beta * xi + gamma) * (ev["b"] + beta * k1 * xi + gamma) % R * (ev["c"] + beta * k2 * xi + gamma) % R * alpha % R
    e3 = (ev["a"] + beta * ev["s1"] + gamma) * (ev["b"] + beta * ev["s2"] + gamma) % R * beta % R * ev["zw"] % R * alpha % R
    xim = pow(xi, n, R)
    l1 = (xim - 1) * pow((xi - 1) * n, -1, R) % R
    e4 = l1 * alpha % R * alpha % R
    coefz, coefs3 = (e2 + e4) % R, e3
    pol_r = []
    for i in range(n + 3):
        v = coefz * pol_z[i] % R
        if i < n:
            v = (v + coef_ab * zk["pol_Qm"][i] + ev["a"] * zk["pol_Ql"][i] + ev["b"] * zk["pol_Qr"][i] +
                 ev["c"] * zk["pol_Qo"][i] + zk["pol_Qc"][i] - coefs3 * zk["pol_S3"][i]) % R
        pol_r.append(v)
    ev["r"] = _eval_pol(pol_r, xi)
    # round 5
    v1 = hash_to_fr(b"".join(fr_be(ev[k]) for k in ("a", "b", "c", "s1", "s2", "zw", "r")))
    v = [0, v1]
    for i in range(2, 7):
        v.append(v[i - 1] * v1 % R)
    xi2m = xim * xim % R
    pol_wxi = []
    for i in range(n + 6):
        x = xi2m * pol_t[2 * n + i] % R
        if i < n + 3:
            x = (x + v[1] * pol_r[i]) % R
        if i < n + 2:
            x = (x + v[2] * pol_a[i] + v[3] * pol_b[i] + v[4] * pol_c[i]) % R
        if i < n:
            x = (x + pol_t[i] + xim * pol_t[n + i] + v[5] * zk["pol_S1"][i] + v[6] * zk["pol_S2"][i]) % R
        pol_wxi.append(x)
    pol_wxi[0] = (pol_wxi[0] - ev["t"] - v[1] * ev["r"] - v[2] * ev["a"] - v[3] * ev["b"] - v[4] * ev["c"] -
                  v[5] * ev["s1"] - v[6] * ev["s2"]) % R
    pol_wxi = _div_pol1(pol_wxi, xi)
    proof["Wxi"] = exp_tau(pol_wxi)
    pol_wxiw = list(pol_z[:n + 3])
    pol_wxiw[0] = (pol_wxiw[0] - ev["zw"]) % R
    pol_wxiw = _div_pol1(pol_wxiw, xi * w1 % R)
    proof["Wxiw"] = exp_tau(pol_wxiw)
    for k in ("a", "b", "c", "s1", "s2", "zw", "r"):
        proof["eval_" + k] = ev[k]
    return proof, [witness[i] % R for i in range(1, zk["nPublic"] + 1)]


# ------------------------------------------------------------------ verifier (plonk_verify.js)
def verify(vk, public, proof):
    if len(public) != vk["nPublic"]:
        return False
    for k in ("A", "B", "C", "Z", "T1", "T2", "T3", "Wxi", "Wxiw"):
        if proof[k] is not None and not G1.on_curve(proof[k]):
            return False
    n = 1 << vk["power"]
    w1 = fr_root(vk["power"])
    k1, k2 = vk["k1"], vk["k2"]
    beta = hash_to_fr(g1_be(proof["A"]) + g1_be(proof["B"]) + g1_be(proof["C"]))
    gamma = hash_to_fr(fr_be(beta))
    alpha = hash_to_fr(g1_be(proof["Z"]))
    xi = hash_to_fr(g1_be(proof["T1"]) + g1_be(proof["T2"]) + g1_be(proof["T3"]))
    a, b, c, s1, s2, zw, r = (proof["eval_" + k] % R for k in ("a", "b", "c", "s1", "s2", "zw", "r"))
    v1 = hash_to_fr(b"".join(fr_be(x) for x in (a, b, c, s1, s2, zw, r)))
    v = [0, v1]
    for i in range(2, 7):
        v.append(v[i - 1] * v1 % R)
    u = hash_to_fr(g1_be(proof["Wxi"]) + g1_be(proof["Wxiw"]))
    xin = pow(xi, n, R)
    zh = (xin - 1) % R
    if zh == 0:
        return False
    Lg = []
    w = 1
    for _ in range(max(1, vk["nPublic"])):
        Lg.append(w * zh % R * pow(n * (xi - w), -1, R) % R)
        w = w * w1 % R
    pl = 0
    for j, pub in enumerate(public):
        pl = (pl - pub * Lg[j]) % R
    t = (r + pl - (a + beta * s1 + gamma) * (b + beta * s2 + gamma) % R * (c + gamma) % R * zw % R * alpha - Lg[0] * alpha * alpha) % R
    t = t * pow(zh, -1, R) % R
    coefz = ((a + beta * xi + gamma) * (b + beta * k1 * xi + gamma) % R * (c + beta * k2 * xi + gamma) % R * alpha + Lg[0] * alpha * alpha) % R
    coefs3 = (a + beta * s1 + gamma) * (b + beta * s2 + gamma) % R * beta % R * zw % R * alpha % R
    # F = [T] + v1 [R] + v2 [A] + v3 [B] + v4 [C] + v5 [S1] + v6 [S2] + u [Z]
    bases = [proof["T1"], proof["T2"], proof["T3"], vk["Qm"], vk["Ql"], vk["Qr"], vk["Qo"], vk["Qc"], vk["S3"], proof["Z"],
             proof["A"], proof["B"], proof["C"], vk["S1"], vk["S2"]]
    scal = [1, xin, xin * xin % R, v[1] * a * b % R, v[1] * a % R, v[1] * b % R, v[1] * c % R, v[1], (-v[1] * coefs3) % R,
            (v[1] * coefz + u) % R, v[2], v[3], v[4], v[5], v[6]]
    pts = [(P, s) for P, s in zip(bases, scal) if P is not None]
    F = G1.msm([P for P, _ in pts], [s for _, s in pts])
    e = (t + v[1] * r + v[2] * a + v[3] * b + v[4] * c + v[5] * s1 + v[6] * s2 + u * zw) % R
    E = G1.mul(G1_GEN, e)
    lhs = G1.add(proof["Wxi"], G1.mul(proof["Wxiw"], u))
    rhs = G1.add(G1.add(G1.mul(proof["Wxi"], xi), G1.mul(proof["Wxiw"], u * xi % R * w1 % R)), G1.add(F, G1.neg(E)))
    return pairing_product_is_one([(lhs, vk["X_2"]), (G1.neg(rhs), G2_GEN)])


# ------------------------------------------------------------------ PLONK .zkey (snarkjs 0.4.12 zkey_utils.js writePlonk / readHeaderPlonk)
def write_zkey(zk, with_lagrange=True):
    """Sections: 1 protocol id (2), 2 header, 3 additions, 4-6 the A/B/C signal maps, 7-11 Qm Ql Qr Qo Qc, 12 sigma 1-3,
    13 Lagrange polynomials of the public inputs, 14 powers of tau; a polynomial = N coefficients + 4N evaluations;
    field elements as Montgomery residues, points affine Montgomery.  with_lagrange=False writes an EMPTY section 13
    (a prover that derives the public-input polynomial by NTT does not read it; 513 public signals at N = 2^22 would
    be 344 GB)."""
    import struct
    from bn254 import Q, RR
    from formats import le, g1_to_lem, g2_to_lem, write_binfile, N8

    def fr(x):
        return le(x * RR % R)

    def poly(c, e4):
        return b"".join(fr(x) for x in c) + b"".join(fr(x) for x in e4)
    n = zk["domainSize"]
    s2 = (struct.pack("<I", N8) + le(Q) + struct.pack("<I", N8) + le(R) +
          struct.pack("<IIIII", zk["nVars"], zk["nPublic"], n, zk["nAdditions"], zk["nConstraints"]) +
          fr(zk["k1"]) + fr(zk["k2"]) + b"".join(g1_to_lem(zk[k]) for k in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3")) +
          g2_to_lem(zk["X_2"]))
    s3 = b"".join(struct.pack("<II", s1, s2_) + fr(f1) + fr(f2) for s1, s2_, f1, f2 in zk["additions"])
    nc = zk["nConstraints"]
    maps = [b"".join(struct.pack("<I", s) for s in zk["maps"][c][:nc]) for c in range(3)]
    secs = [(1, struct.pack("<I", 2)), (2, s2), (3, s3), (4, maps[0]), (5, maps[1]), (6, maps[2])]
    for sid, name in zip(range(7, 12), ("Qm", "Ql", "Qr", "Qo", "Qc")):
        secs.append((sid, poly(zk["pol_" + name], zk["ext_" + name])))
    secs.append((12, b"".join(poly(zk["pol_S%d" % k], zk["ext_S%d" % k]) for k in (1, 2, 3))))
    secs.append((13, b"".join(poly(c, e4) for c, e4 in zk["lagrange"]) if with_lagrange else b""))
    secs.append((14, b"".join(g1_to_lem(P) for P in zk["srs"])))
    return write_binfile("zkey", 1, secs)


def proof_obj(proof):
    """The object `snarkjs plonk prove` stringifies (key order of plonk_prove.js)."""
    def g1(P):
        return ["0", "1", "0"] if P is None else [str(P[0]), str(P[1]), "1"]
    o = {}
    for k in ("A", "B", "C", "Z", "T1", "T2", "T3"):
        o[k] = g1(proof[k])
    for k in ("a", "b", "c", "s1", "s2", "zw", "r"):
        o["eval_" + k] = str(proof["eval_" + k])
    o["Wxi"], o["Wxiw"] = g1(proof["Wxi"]), g1(proof["Wxiw"])
    o["protocol"], o["curve"] = "plonk", "bn128"
    return o


def proof_from_obj(o):
    def g1(t):
        return None if str(t[2]) == "0" else (int(t[0]), int(t[1]))
    p = {k: g1(o[k]) for k in ("A", "B", "C", "Z", "T1", "T2", "T3", "Wxi", "Wxiw")}
    for k in ("a", "b", "c", "s1", "s2", "zw", "r"):
        p["eval_" + k] = int(o["eval_" + k])
    return p


def vkey_from_zkey(buf):
    """The verification key `snarkjs zkey export verificationkey` reads from a PLONK .zkey header (section 2)."""
    import struct
    from bn254 import RR
    from formats import read_binfile, section, g1_from_lem, g2_from_lem, from_le
    secs = read_binfile(buf, "zkey", 2, "zkey")
    h = section(buf, secs, 2)
    pos = 72
    n_vars, n_public, n, n_add, n_cons = struct.unpack_from("<IIIII", h, pos)
    pos += 20
    rinv = pow(RR, -1, R)
    k1 = from_le(h[pos:pos + 32]) * rinv % R
    k2 = from_le(h[pos + 32:pos + 64]) * rinv % R
    pos += 64
    vk = {"protocol": "plonk", "nPublic": n_public, "power": n.bit_length() - 1, "k1": k1, "k2": k2}
    for name in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
        vk[name] = g1_from_lem(h[pos:pos + 64])
        pos += 64
    vk["X_2"] = g2_from_lem(h[pos:pos + 128])
    return vk


def write_ptau(power, tau):
    """A powers-of-tau file as snarkjs's powersoftau_utils.js lays it out (.ptau v1): section 1 = n8, q, power,
    ceremonyPower; 2 = [tau^i]G1, i < 2^(power+1) - 1; 3 = [tau^i]G2, i < 2^power; 4-6 the alpha/beta points (dummies
    here: the PLONK setup does not read them); 7 = contributions (none).  Test fixture generator: tau is known."""
    import struct
    from bn254 import Q
    from formats import le, g1_to_lem, g2_to_lem, write_binfile, N8
    n = 1 << power
    s1 = struct.pack("<I", N8) + le(Q) + struct.pack("<II", power, power)
    pw = [pow(tau, i, R) for i in range(2 * n - 1)]
    s2 = b"".join(g1_to_lem(P) for P in G1.gen_mul_many(pw))
    s3 = b"".join(g2_to_lem(P) for P in G2.gen_mul_many(pw[:n]))
    s4 = b"".join(g1_to_lem(G1_GEN) for _ in range(n))
    s5 = s4
    s6 = g2_to_lem(G2_GEN)
    return write_binfile("ptau", 1, [(1, s1), (2, s2), (3, s3), (4, s4), (5, s5), (6, s6), (7, struct.pack("<I", 0))])

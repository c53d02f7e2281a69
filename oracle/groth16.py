"""ORACLE (test infrastructure) -- Groth16 prove / trapdoor setup / verify on Python ints.

PARITY UNPINNED against the reference (no prover source or golden proof in /root/reference;
the only snarkjs call sites are the PLONK CLI lines /root/reference/Makefile:30-33).  This file
restates the *published* snarkjs 0.4.12 `groth16_prove.js` pipeline (yarn.lock:987-1001) as
recorded in SURVEY.md section 3.3 and App. C:

  buildABC1 -> 3 x (Fr.ifft, batchApplyKey(inc = w_{2N}), Fr.fft) -> joinABC -> 5 x multiExpAffine
  -> blinding with (r, s) -> affine -> decimal JSON.

Independent pins carried instead: the trapdoor known-answer oracle `expected_proof`
(App. C.4, needs no MSM/NTT at all) and the pairing check `verify` (bn254.pairing_product_is_one).
"""
from bn254 import R, G1, G2, G1_GEN, G2_GEN, fr_root, pairing_product_is_one
from synth import Xoshiro


def log2_exact(n):
    p = n.bit_length() - 1
    assert 1 << p == n
    return p


# ------------------------------------------------------------------ NTT (natural order in/out)
def ntt(vals, inverse=False):
    """DFT over Fr: out[i] = sum_j vals[j] * w^(ij), w = Fr.w[log2 N]; the inverse includes 1/N
    (ffjavascript Fr.fft / Fr.ifft semantics [EXT])."""
    n = len(vals)
    power = log2_exact(n)
    w = fr_root(power)
    if inverse:
        w = pow(w, -1, R)
    a = list(vals)
    # bit reversal
    j = 0
    for i in range(1, n):
        bit = n >> 1
        while j & bit:
            j ^= bit
            bit >>= 1
        j |= bit
        if i < j:
            a[i], a[j] = a[j], a[i]
    length = 2
    while length <= n:
        wl = pow(w, n // length, R)
        half = length >> 1
        for start in range(0, n, length):
            t = 1
            for k in range(start, start + half):
                u = a[k]
                v = a[k + half] * t % R
                a[k] = (u + v) % R
                a[k + half] = (u - v) % R
                t = t * wl % R
        length <<= 1
    if inverse:
        ninv = pow(n, -1, R)
        a = [x * ninv % R for x in a]
    return a


# ------------------------------------------------------------------ prover
def build_abc(zk, w):
    """buildABC1: A_T, B_T from the section-4 records, C_T = A_T o B_T."""
    N = zk["domainSize"]
    a = [0] * N
    b = [0] * N
    for (m, c, s, v) in zk["coefs"]:
        if m == 0:
            a[c] = (a[c] + v * w[s]) % R
        else:
            b[c] = (b[c] + v * w[s]) % R
    cc = [x * y % R for x, y in zip(a, b)]
    return a, b, cc


def h_scalars(zk, w):
    """P_i = A(w_2N^(2i+1)) * B(...) - C(...), i < N (the H-MSM scalars, standard form)."""
    N = zk["domainSize"]
    power = log2_exact(N)
    inc = fr_root(power + 1) if power < 28 else 25  # Fr.shift = nqr^2 when power == Fr.s [EXT]
    outs = []
    for ev in build_abc(zk, w):
        coef = ntt(ev, inverse=True)
        t = 1
        sh = []
        for x in coef:
            sh.append(x * t % R)
            t = t * inc % R
        outs.append(ntt(sh))
    a, b, c = outs
    return [(x * y - z) % R for x, y, z in zip(a, b, c)]


def prove(zk, w, r, s):
    """Returns (A, B, C) affine with A in G1, B in G2, C in G1, and the public signals."""
    n, p = zk["nVars"], zk["nPublic"]
    if len(w) != n:
        raise ValueError(f"Invalid witness length. Circuit: {n}, witness: {len(w)}")
    P = h_scalars(zk, w)
    j = G1.to_jac
    pi_a = G1.jadd(j(G1.msm(zk["A"], w)), j(zk["alpha1"]))
    pi_a = G1.jadd(pi_a, G1.jmul(zk["delta1"], r))
    pi_b = G2.jadd(G2.to_jac(G2.msm(zk["B2"], w)), G2.to_jac(zk["beta2"]))
    pi_b = G2.jadd(pi_b, G2.jmul(zk["delta2"], s))
    pib1 = G1.jadd(j(G1.msm(zk["B1"], w)), j(zk["beta1"]))
    pib1 = G1.jadd(pib1, G1.jmul(zk["delta1"], s))
    pi_c = G1.jadd(j(G1.msm(zk["C"], w[p + 1:])), j(G1.msm(zk["H"], P)))
    A_aff = G1.to_affine(pi_a)
    pi_c = G1.jadd(pi_c, G1.jmul(A_aff, s))
    pi_c = G1.jadd(pi_c, G1.jmul(G1.to_affine(pib1), r))
    pi_c = G1.jadd(pi_c, G1.jmul(zk["delta1"], (-(r * s)) % R))
    return (A_aff, G2.to_affine(pi_b), G1.to_affine(pi_c)), list(w[1:p + 1])


# ------------------------------------------------------------------ trapdoor setup (App. C.4)
def lagrange_at(N, tau):
    """L_c(tau), c < N, for the domain {w_N^c}."""
    power = log2_exact(N)
    w = fr_root(power)
    zt = (pow(tau, N, R) - 1) % R
    ninv = pow(N, -1, R)
    out = []
    wc = 1
    for _ in range(N):
        out.append(zt * ninv % R * wc % R * pow((tau - wc) % R, -1, R) % R)
        wc = wc * w % R
    return out


def trapdoor(seed):
    rng = Xoshiro(seed)
    vals = []
    while len(vals) < 5:
        v = rng.rand_fr()
        if v:
            vals.append(v)
    return dict(zip(("tau", "alpha", "beta", "gamma", "delta"), vals))


def setup(n, p, rows, td):
    """Test-only Groth16 setup with a known trapdoor -> (zkey dict, scalar-side secrets).
    Layout and H basis as SURVEY App. A.3 / C.3."""
    m = len(rows)
    N = 1
    while N < m + p + 1:
        N <<= 1
    tau, alpha, beta, gamma, delta = (td[k] for k in ("tau", "alpha", "beta", "gamma", "delta"))
    L = lagrange_at(N, tau)
    u = [0] * n
    v = [0] * n
    t = [0] * n
    coefs = []
    for c, (A, B, C) in enumerate(rows):
        for s, cf in A:
            coefs.append((0, c, s, cf)); u[s] = (u[s] + cf * L[c]) % R
        for s, cf in B:
            coefs.append((1, c, s, cf)); v[s] = (v[s] + cf * L[c]) % R
        for s, cf in C:
            t[s] = (t[s] + cf * L[c]) % R
    for i in range(p + 1):                                   # public-input binding rows
        coefs.append((0, m + i, i, 1)); u[i] = (u[i] + L[m + i]) % R
    ginv, dinv = pow(gamma, -1, R), pow(delta, -1, R)
    kk = [(beta * u[i] + alpha * v[i] + t[i]) % R for i in range(n)]
    # H_i = [L^(2N)_{2i+1}(tau) / delta]
    L2 = lagrange_at(2 * N, tau)
    hs = [L2[2 * i + 1] * dinv % R for i in range(N)]
    zk = {
        "nVars": n, "nPublic": p, "domainSize": N,
        "alpha1": G1.mul(G1_GEN, alpha), "beta1": G1.mul(G1_GEN, beta),
        "beta2": G2.mul(G2_GEN, beta), "gamma2": G2.mul(G2_GEN, gamma),
        "delta1": G1.mul(G1_GEN, delta), "delta2": G2.mul(G2_GEN, delta),
        "IC": G1.gen_mul_many([kk[i] * ginv % R for i in range(p + 1)]),
        "coefs": coefs,
        "A": G1.gen_mul_many(u), "B1": G1.gen_mul_many(v), "B2": G2.gen_mul_many(v),
        "C": G1.gen_mul_many([kk[i] * dinv % R for i in range(p + 1, n)]),
        "H": G1.gen_mul_many(hs),
    }
    return zk, {"u": u, "v": v, "t": t, **td}


def expected_proof_scalars(sec, p, w, r, s):
    """(a, b, c) in Fr with proof == ([a]G1, [b]G2, [c]G1)  (SURVEY App. C.4)."""
    u, v, t = sec["u"], sec["v"], sec["t"]
    alpha, beta, delta = sec["alpha"], sec["beta"], sec["delta"]
    wu = sum(x * y for x, y in zip(w, u)) % R
    wv = sum(x * y for x, y in zip(w, v)) % R
    wt = sum(x * y for x, y in zip(w, t)) % R
    a = (alpha + wu + r * delta) % R
    b = (beta + wv + s * delta) % R
    priv = sum(w[i] * (beta * u[i] + alpha * v[i] + t[i]) for i in range(p + 1, len(w))) % R
    c = ((priv + wu * wv - wt) * pow(delta, -1, R) + s * a + r * b - r * s * delta) % R
    return a, b, c


def expected_proof(sec, p, w, r, s):
    a, b, c = expected_proof_scalars(sec, p, w, r, s)
    return G1.mul(G1_GEN, a), G2.mul(G2_GEN, b), G1.mul(G1_GEN, c)


# ------------------------------------------------------------------ verify
def verify(vk, public, proof):
    """e(-A,B) e(alpha,beta) e(vk_x,gamma) e(C,delta) == 1; vk = dict with alpha1, beta2, gamma2,
    delta2, IC (affine points); proof = (A, B, C) affine."""
    A, B, C = proof
    if len(public) + 1 != len(vk["IC"]):
        return False
    for P in (A, C):
        if P is not None and not G1.on_curve(P):
            return False
    if B is not None and not G2.on_curve(B):
        return False
    vkx = G1.jadd(G1.to_jac(vk["IC"][0]),
                  G1.to_jac(G1.msm(vk["IC"][1:], [x % R for x in public]))) if public \
        else G1.to_jac(vk["IC"][0])
    vkx = G1.to_affine(vkx)
    return pairing_product_is_one([(G1.neg(A), B), (vk["alpha1"], vk["beta2"]),
                                   (vkx, vk["gamma2"]), (C, vk["delta2"])])

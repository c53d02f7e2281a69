"""ORACLE (test infrastructure, not product code) -- BN254 ("bn128") arithmetic on Python ints.

PARITY UNPINNED against the reference: the reference repo (/root/reference) holds no
prover source, golden proof, zkey or wtns.  The algorithm lives in
un-vendored npm packages pinned by /root/reference/yarn.lock:
  snarkjs 0.4.12 (yarn.lock:987-1001), ffjavascript 0.2.48 (yarn.lock:408-416),
  wasmcurves 0.1.0 (yarn.lock:1132-1138).
This file restates their *published* mathematics (curve constants, Montgomery radix,
2-adic root of unity, tower for the pairing) from SURVEY.md App. B, every constant of
which is re-checked numerically by tests/test_oracle_constants.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""

# ---------------------------------------------------------------- constants
Q = 21888242871839275222246405745257275088696311157297823662689037894645226208583  # base field
R = 21888242871839275222246405745257275088548364400416034343698204186575808495617  # scalar field
MONT = 1 << 256                      # Montgomery radix used by wasmcurves build_f1m.js [EXT]
RQ = MONT % Q
RR = MONT % R
R2Q = (MONT * MONT) % Q
R2R = (MONT * MONT) % R
FR_S = 28                            # 2-adicity of r-1
FR_T = (R - 1) >> FR_S
FR_NQR = 5                           # smallest quadratic non-residue (ffjavascript nqr) [EXT]
FR_W28 = pow(FR_NQR, FR_T, R)        # primitive 2^28-th root of unity


def fr_root(power):
    """omega_{2^power} = Fr.w[power]; Fr.w[i] = Fr.w[i+1]^2 (ffjavascript F1Field) [EXT]."""
    assert 0 <= power <= FR_S
    return pow(FR_W28, 1 << (FR_S - power), R)


G1_GEN = (1, 2)
G1_B = 3
# Fq2 = Fq[u]/(u^2+1); element = (c0, c1)
G2_B = (19485874751759354771024239261021720505790618469301721065564631296452457478373,
        266929791119991161246907387137283842545076965332900288569378510910307636690)
G2_GEN = ((10857046999023057135944570762232829481370756359578518086990519993285655852781,
           11559732032986387107991004021392285783925812861821192530917403151452391805634),
          (8495653923123431417604973247489272438418190587263600148770280649306958101930,
           4082367875863433681332203403145435568316851327593401208105741076214120093531))


# ---------------------------------------------------------------- Fq2
def f2_add(a, b): return ((a[0] + b[0]) % Q, (a[1] + b[1]) % Q)
def f2_sub(a, b): return ((a[0] - b[0]) % Q, (a[1] - b[1]) % Q)
def f2_neg(a): return ((-a[0]) % Q, (-a[1]) % Q)
def f2_mul(a, b):
    return ((a[0] * b[0] - a[1] * b[1]) % Q, (a[0] * b[1] + a[1] * b[0]) % Q)
def f2_sqr(a): return f2_mul(a, a)
def f2_scale(a, k): return ((a[0] * k) % Q, (a[1] * k) % Q)
def f2_inv(a):
    d = pow((a[0] * a[0] + a[1] * a[1]) % Q, -1, Q)
    return ((a[0] * d) % Q, (-a[1] * d) % Q)
F2_ZERO = (0, 0)
F2_ONE = (1, 0)


class _Fq:
    """Field ops bundle so the generic curve code below serves G1 (ints) and G2 (pairs)."""
    zero, one = 0, 1
    add = staticmethod(lambda a, b: (a + b) % Q)
    sub = staticmethod(lambda a, b: (a - b) % Q)
    mul = staticmethod(lambda a, b: (a * b) % Q)
    neg = staticmethod(lambda a: (-a) % Q)
    inv = staticmethod(lambda a: pow(a, -1, Q))
    is_zero = staticmethod(lambda a: a % Q == 0)


class _Fq2:
    zero, one = F2_ZERO, F2_ONE
    add, sub, mul, neg, inv = map(staticmethod, (f2_add, f2_sub, f2_mul, f2_neg, f2_inv))
    is_zero = staticmethod(lambda a: a[0] % Q == 0 and a[1] % Q == 0)


# ---------------------------------------------------------------- short Weierstrass, a = 0
# Points: None = infinity, else affine (x, y).  Jacobian (X, Y, Z) internally.
class Curve:
    def __init__(self, F, b, gen):
        self.F, self.b, self.gen = F, b, gen

    def on_curve(self, P):
        if P is None:
            return True
        F = self.F
        x, y = P
        return F.sub(F.mul(y, y), F.add(F.mul(F.mul(x, x), x), self.b)) == F.zero

    def neg(self, P):
        return None if P is None else (P[0], self.F.neg(P[1]))

    # Jacobian arithmetic
    def to_jac(self, P):
        return (self.F.one, self.F.one, self.F.zero) if P is None else (P[0], P[1], self.F.one)

    def jdbl(self, P):
        F = self.F
        X, Y, Z = P
        if F.is_zero(Z) or F.is_zero(Y):
            return (F.one, F.one, F.zero)
        A = F.mul(X, X); B = F.mul(Y, Y); C = F.mul(B, B)
        t = F.add(X, B)
        D = F.sub(F.sub(F.mul(t, t), A), C); D = F.add(D, D)
        E = F.add(F.add(A, A), A)
        Fq_ = F.mul(E, E)
        X3 = F.sub(Fq_, F.add(D, D))
        C8 = F.add(C, C); C8 = F.add(C8, C8); C8 = F.add(C8, C8)
        Y3 = F.sub(F.mul(E, F.sub(D, X3)), C8)
        Z3 = F.mul(F.add(Y, Y), Z)
        return (X3, Y3, Z3)

    def jadd(self, P, Qp):
        F = self.F
        X1, Y1, Z1 = P
        X2, Y2, Z2 = Qp
        if F.is_zero(Z1):
            return Qp
        if F.is_zero(Z2):
            return P
        Z1Z1 = F.mul(Z1, Z1); Z2Z2 = F.mul(Z2, Z2)
        U1 = F.mul(X1, Z2Z2); U2 = F.mul(X2, Z1Z1)
        S1 = F.mul(F.mul(Y1, Z2), Z2Z2); S2 = F.mul(F.mul(Y2, Z1), Z1Z1)
        H = F.sub(U2, U1); Rr = F.sub(S2, S1)
        if F.is_zero(H):
            if F.is_zero(Rr):
                return self.jdbl(P)
            return (F.one, F.one, F.zero)
        HH = F.mul(H, H); HHH = F.mul(H, HH); V = F.mul(U1, HH)
        X3 = F.sub(F.sub(F.mul(Rr, Rr), HHH), F.add(V, V))
        Y3 = F.sub(F.mul(Rr, F.sub(V, X3)), F.mul(S1, HHH))
        Z3 = F.mul(F.mul(Z1, Z2), H)
        return (X3, Y3, Z3)

    def to_affine(self, P):
        F = self.F
        X, Y, Z = P
        if F.is_zero(Z):
            return None
        zi = F.inv(Z); zi2 = F.mul(zi, zi)
        return (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))

    def add(self, P, Qp):
        return self.to_affine(self.jadd(self.to_jac(P), self.to_jac(Qp)))

    def jmul(self, P, k):
        """k * P (affine in, Jacobian out), plain double-and-add."""
        F = self.F
        acc = (F.one, F.one, F.zero)
        if P is None or k == 0:
            return acc
        J = self.to_jac(P)
        for bit in bin(k)[2:]:
            acc = self.jdbl(acc)
            if bit == '1':
                acc = self.jadd(acc, J)
        return acc

    def mul(self, P, k):
        return self.to_affine(self.jmul(P, k % R))

    def batch_to_affine(self, Js):
        """Montgomery batch inversion of the Z coordinates."""
        F = self.F
        n = len(Js)
        pref = [None] * n
        acc = F.one
        for i, (_, _, Z) in enumerate(Js):
            pref[i] = acc
            if not F.is_zero(Z):
                acc = F.mul(acc, Z)
        inv = F.inv(acc)
        out = [None] * n
        for i in range(n - 1, -1, -1):
            X, Y, Z = Js[i]
            if F.is_zero(Z):
                continue
            zi = F.mul(inv, pref[i])
            inv = F.mul(inv, Z)
            zi2 = F.mul(zi, zi)
            out[i] = (F.mul(X, zi2), F.mul(Y, F.mul(zi2, zi)))
        return out

    def fixed_base_table(self, w=4):
        """table[j][d] = d * 2^(w*j) * G (Jacobian) for the windowed generator multiply."""
        nwin = (254 + w - 1) // w
        tbl = []
        base = self.to_jac(self.gen)
        for _ in range(nwin):
            row = [(self.F.one, self.F.one, self.F.zero)]
            for d in range(1, 1 << w):
                row.append(self.jadd(row[-1], base))
            aff = self.batch_to_affine(row)
            tbl.append([self.to_jac(p) for p in aff])
            for _ in range(w):
                base = self.jdbl(base)
        return tbl

    def gen_mul_many(self, ks, w=4):
        """[k]G for many k (affine list) -- the trapdoor setup's fixed-base multiply."""
        if not hasattr(self, "_tbl"):
            self._tbl = self.fixed_base_table(w)
            self._w = w
        w = self._w
        mask = (1 << w) - 1
        Js = []
        for k in ks:
            k %= R
            acc = (self.F.one, self.F.one, self.F.zero)
            j = 0
            while k:
                d = k & mask
                if d:
                    acc = self.jadd(acc, self._tbl[j][d])
                k >>= w
                j += 1
            Js.append(acc)
        return self.batch_to_affine(Js)

    def msm(self, bases, scalars, c=8):
        """Sum_i scalars[i]*bases[i]; simple Pippenger (unsigned c-bit windows); affine out."""
        F = self.F
        inf = (F.one, F.one, F.zero)
        assert len(bases) == len(scalars)
        nwin = (254 + c - 1) // c
        total = inf
        for wdx in range(nwin - 1, -1, -1):
            for _ in range(c):
                total = self.jdbl(total)
            buckets = [inf] * (1 << c)
            sh = wdx * c
            for P, s in zip(bases, scalars):
                if P is None:
                    continue
                d = (s >> sh) & ((1 << c) - 1)
                if d:
                    buckets[d] = self.jadd(buckets[d], self.to_jac(P))
            run = inf
            acc = inf
            for d in range((1 << c) - 1, 0, -1):
                run = self.jadd(run, buckets[d])
                acc = self.jadd(acc, run)
            total = self.jadd(total, acc)
        return self.to_affine(total)


G1 = Curve(_Fq, G1_B, G1_GEN)
G2 = Curve(_Fq2, G2_B, G2_GEN)


# ---------------------------------------------------------------- pairing (verifier only)
# Fq12 = Fq[w]/(w^12 - 18 w^6 + 82): the direct degree-12 form of the tower
# Fq2[v]/(v^3-(9+u)), [w]/(w^2-v).  Slow and simple on purpose.
_F12_MOD = [82, 0, 0, 0, 0, 0, -18, 0, 0, 0, 0, 0]  # w^12 = 18 w^6 - 82


def f12_mul(a, b):
    t = [0] * 23
    for i, ai in enumerate(a):
        if ai:
            for j, bj in enumerate(b):
                t[i + j] += ai * bj
    for k in range(22, 11, -1):
        top = t[k]
        if top:
            t[k] = 0
            t[k - 6] += 18 * top
            t[k - 12] -= 82 * top
    return [x % Q for x in t[:12]]


F12_ONE = [1] + [0] * 11


def f12_pow(a, e):
    r = F12_ONE
    for bit in bin(e)[2:]:
        r = f12_mul(r, r)
        if bit == '1':
            r = f12_mul(r, a)
    return r


def _poly_deg(p):
    d = len(p) - 1
    while d and p[d] == 0:
        d -= 1
    return d


def f12_inv(a):
    """Extended Euclid over Fq[w] against the modulus polynomial."""
    lm, hm = [1] + [0] * 12, [0] * 13
    low = list(a) + [0]
    high = [x % Q for x in _F12_MOD] + [1]
    while _poly_deg(low):
        # r = high // low (polynomial rounded division)
        dl, dh = _poly_deg(low), _poly_deg(high)
        temp = list(high)
        o = [0] * 13
        for i in range(dh - dl, -1, -1):
            o[i] = (o[i] + temp[dl + i] * pow(low[dl], -1, Q)) % Q
            for c in range(dl + 1):
                temp[c + i] = (temp[c + i] - o[i] * low[c]) % Q  # o[i] fixed before use
        r = o[:dh - dl + 1] + [0] * (13 - (dh - dl + 1))
        nm = list(hm)
        new = list(high)
        for i in range(13):
            for j in range(13 - i):
                nm[i + j] -= lm[i] * r[j]
                new[i + j] -= low[i] * r[j]
        nm = [x % Q for x in nm]
        new = [x % Q for x in new]
        lm, low, hm, high = nm, new, lm, low
    inv0 = pow(low[0], -1, Q)
    return [(x * inv0) % Q for x in lm[:12]]


def _f12_from_fq(x):
    return [x % Q] + [0] * 11


def _twist(P):
    """G2 affine (Fq2 coords) -> curve over Fq12 (untwist), as in the w^12-18w^6+82 form."""
    (x0, x1), (y0, y1) = P
    xc = [(x0 - 9 * x1) % Q, x1]
    yc = [(y0 - 9 * y1) % Q, y1]
    nx = [xc[0], 0, 0, 0, 0, 0, xc[1], 0, 0, 0, 0, 0]
    ny = [yc[0], 0, 0, 0, 0, 0, yc[1], 0, 0, 0, 0, 0]
    w2 = [0, 0, 1] + [0] * 9
    w3 = [0, 0, 0, 1] + [0] * 8
    return (f12_mul(nx, w2), f12_mul(ny, w3))


def _f12_sub(a, b): return [(x - y) % Q for x, y in zip(a, b)]
def _f12_add(a, b): return [(x + y) % Q for x, y in zip(a, b)]
def _f12_scale(a, k): return [(x * k) % Q for x in a]


def _linefunc(P1, P2, T):
    x1, y1 = P1; x2, y2 = P2; xt, yt = T
    if x1 != x2:
        m = f12_mul(_f12_sub(y2, y1), f12_inv(_f12_sub(x2, x1)))
        return _f12_sub(f12_mul(m, _f12_sub(xt, x1)), _f12_sub(yt, y1))
    if y1 == y2:
        m = f12_mul(_f12_scale(f12_mul(x1, x1), 3), f12_inv(_f12_scale(y1, 2)))
        return _f12_sub(f12_mul(m, _f12_sub(xt, x1)), _f12_sub(yt, y1))
    return _f12_sub(xt, x1)


def _f12_pt_double(P):
    x, y = P
    m = f12_mul(_f12_scale(f12_mul(x, x), 3), f12_inv(_f12_scale(y, 2)))
    nx = _f12_sub(f12_mul(m, m), _f12_scale(x, 2))
    ny = _f12_sub(f12_mul(m, _f12_sub(x, nx)), y)
    return (nx, ny)


def _f12_pt_add(P1, P2):
    x1, y1 = P1; x2, y2 = P2
    if x1 == x2 and y1 == y2:
        return _f12_pt_double(P1)
    m = f12_mul(_f12_sub(y2, y1), f12_inv(_f12_sub(x2, x1)))
    nx = _f12_sub(_f12_sub(f12_mul(m, m), x1), x2)
    ny = _f12_sub(f12_mul(m, _f12_sub(x1, nx)), y1)
    return (nx, ny)


ATE_LOOP = 29793968203157093288          # 6u+2, u = 4965661367192848881
FINAL_EXP = (Q ** 12 - 1) // R


def miller_loop(Q2, P1):
    """Q2: G2 affine, P1: G1 affine (neither infinity).  Returns f before final exponentiation."""
    Qt = _twist(Q2)
    Pt = (_f12_from_fq(P1[0]), _f12_from_fq(P1[1]))
    Rp = Qt
    f = F12_ONE
    for i in range(ATE_LOOP.bit_length() - 2, -1, -1):
        f = f12_mul(f12_mul(f, f), _linefunc(Rp, Rp, Pt))
        Rp = _f12_pt_double(Rp)
        if (ATE_LOOP >> i) & 1:
            f = f12_mul(f, _linefunc(Rp, Qt, Pt))
            Rp = _f12_pt_add(Rp, Qt)
    Q1 = (f12_pow(Qt[0], Q), f12_pow(Qt[1], Q))
    nQ2 = (f12_pow(Q1[0], Q), [(-x) % Q for x in f12_pow(Q1[1], Q)])
    f = f12_mul(f, _linefunc(Rp, Q1, Pt))
    Rp = _f12_pt_add(Rp, Q1)
    f = f12_mul(f, _linefunc(Rp, nQ2, Pt))
    return f


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 for [(G1 affine, G2 affine)]; infinity pairs contribute 1."""
    f = F12_ONE
    for P1, Q2 in pairs:
        if P1 is None or Q2 is None:
            continue
        f = f12_mul(f, miller_loop(Q2, P1))
    return f12_pow(f, FINAL_EXP) == F12_ONE

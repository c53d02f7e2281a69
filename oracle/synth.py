"""ORACLE (test infrastructure) -- shape-matched synthetic R1CS + satisfying witness.

The real nzcp_live R1CS cannot be produced offline (no circom; the circuit's includes are fetched
by curl, /root/reference/Makefile:14-19).  SURVEY.md section 8(d) config 2 prescribes a
synthetic stand-in whose shape follows the circuit: p = 513 public outputs
(/root/reference/test/nzcp.js:41-47), rows <= 4 terms in A / <= 2 in B, B touching ~35 % of the
signals, coefficient mix 80 % +-1 / 15 % < 2^16 / 5 % uniform, witness mix 30 % zero / 30 % one /
8 % < 2^10 / 32 % uniform.

THE GENERATOR SPEC (the product's C++ generator in nzcp-circom_amd/csrc/synth.cpp follows the same
draw order, so both sides produce identical bytes for a given seed):

  rng = xoshiro256** seeded with splitmix64(seed) x4.
  below(k)  = next_u64() % k
  rand_fr() = (u0 | u1<<64 | u2<<128 | u3<<192) mod r   (u0 drawn first)

  witness (stream seed):      w[0]=1; for i in 1..n-1:  if i <= p: (i == p ? 1951416330 : below(2))
                              else u=below(100): u<30 -> 0; u<60 -> 1; u<68 -> below(1024); else rand_fr()
  constraints (stream seed+3): for c in 0..m-1:
        ta = 1+below(4); A terms: ta x (s=below(n), coef())
        tb = 1+below(2); B terms: tb x (s=bsel(), coef())       bsel(): t=below(n); t - t%20 + below(7)
                                                                 (clamped to n-1)
        j1 = below(n); c1 = rand_fr();   C row = { j1: c1, 0: a*b - c1*w[j1] }   (merged if j1 == 0)
  coef(): u=below(100): u<40 -> 1; u<80 -> r-1; u<95 -> 1+below(65535); else rand_fr()
  Duplicate signals inside one row are kept as separate terms (snarkjs sums them at prove time).
"""
from bn254 import R

MASK = (1 << 64) - 1
SEED_NZCP = 0x6E7A6370  # "nzcp"
EXP_EXAMPLE = 1951416330  # 'exp' of the MoH example pass


def _rotl(x, k):
    return ((x << k) | (x >> (64 - k))) & MASK


class Xoshiro:
    def __init__(self, seed):
        z = seed & MASK
        s = []
        for _ in range(4):
            z = (z + 0x9E3779B97F4A7C15) & MASK
            x = z
            x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & MASK
            x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & MASK
            s.append(x ^ (x >> 31))
        self.s = s

    def next(self):
        s = self.s
        res = (_rotl((s[1] * 5) & MASK, 7) * 9) & MASK
        t = (s[1] << 17) & MASK
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = _rotl(s[3], 45)
        return res

    def below(self, k):
        return self.next() % k

    def rand_fr(self):
        u = [self.next() for _ in range(4)]
        return (u[0] | (u[1] << 64) | (u[2] << 128) | (u[3] << 192)) % R


def gen_witness(n, p, seed):
    rng = Xoshiro(seed)
    w = [1] + [0] * (n - 1)
    for i in range(1, n):
        if i <= p:
            w[i] = EXP_EXAMPLE if i == p else rng.below(2)
        else:
            u = rng.below(100)
            if u < 30:
                w[i] = 0
            elif u < 60:
                w[i] = 1
            elif u < 68:
                w[i] = rng.below(1024)
            else:
                w[i] = rng.rand_fr()
    return w


def _coef(rng):
    u = rng.below(100)
    if u < 40:
        return 1
    if u < 80:
        return R - 1
    if u < 95:
        return 1 + rng.below(65535)
    return rng.rand_fr()


def gen_circuit(n, p, m, seed, w):
    """Returns rows: list of (A_terms, B_terms, C_terms), terms = [(signal, coef)].
    The circuit is built around witness w (seed stream `seed`), constraints from `seed+3`."""
    rng = Xoshiro(seed + 3)
    rows = []
    for _ in range(m):
        ta = 1 + rng.below(4)
        A = []
        for _ in range(ta):
            s = rng.below(n)
            A.append((s, _coef(rng)))
        tb = 1 + rng.below(2)
        B = []
        for _ in range(tb):
            t = rng.below(n)
            s = t - t % 20 + rng.below(7)
            if s > n - 1:
                s = n - 1
            B.append((s, _coef(rng)))
        j1 = rng.below(n)
        c1 = rng.rand_fr()
        a = sum(cf * w[s] for s, cf in A) % R
        b = sum(cf * w[s] for s, cf in B) % R
        c0 = (a * b - c1 * w[j1]) % R
        C = [(0, (c0 + c1) % R)] if j1 == 0 else [(j1, c1), (0, c0)]
        rows.append((A, B, C))
    return rows


def check_r1cs(rows, w):
    for A, B, C in rows:
        a = sum(cf * w[s] for s, cf in A) % R
        b = sum(cf * w[s] for s, cf in B) % R
        c = sum(cf * w[s] for s, cf in C) % R
        if (a * b - c) % R:
            return False
    return True

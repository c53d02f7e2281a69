"""ORACLE (test infrastructure) -- shape-matched synthetic R1CS + satisfying witnesses.

The real nzcp_live R1CS cannot be produced offline (no circom; the circuit's includes are fetched
by curl, /root/reference/Makefile:14-19).  SURVEY.md section 8(d) config 2 prescribes a
synthetic stand-in whose shape follows the circuit: p = 513 public outputs
(/root/reference/test/nzcp.js:41-47), rows <= 4 terms in A, B touching ~35 % of the signals,
coefficient mix 80 % +-1 / 15 % < 2^16 / 5 % uniform, witness mix 30 % zero / 30 % one /
8 % < 2^10 / 32 % uniform.  Config 3 needs MANY satisfying witnesses for ONE
circuit, so the circuit is built from two constraint kinds that are satisfiable for every choice
of the free wires:
  identity:  (ca*(x - y)) * (cb*(x + y - 1)) = 0          for bit wires x, y   (SHA-like booleans)
  slack:     (A.w) * (B.w) = c1*w[j1] + w[sw]              sw = a fresh full-width "slack" wire
                                                           (IsZero-inverse-like, solved last)

THE GENERATOR SPEC (the product's C++ generator nzcp-circom_amd/csrc/synth.cpp follows the same
draw order, so both sides produce identical bytes for a given seed):

  rng = xoshiro256** seeded with splitmix64(seed) x4;  below(k) = next_u64() % k
  rand_fr() = (u0 | u1<<64 | u2<<128 | u3<<192) mod r   (u0 drawn first)
  coef(): u=below(100): u<40 -> 1; u<80 -> r-1; u<95 -> 1+below(65535); else rand_fr()

  classes (stream seed):   cls[0]=CONST; i in 1..n-1: i<=p -> PUB; else u=below(100):
                           u<60 -> BIT; u<68 -> SMALL; else SLACK
  constraints (stream seed+3):
     slack[] = SLACK wires in index order (rank = position); bits[] = wires i with
     (cls PUB and i<p, or cls BIT) and i%20 < 7, in index order; if bits[] is empty it is [0].
     next = 0
     for c in 0..m-1:
        u = below(100); rem_s = len(slack)-next
        if rem_s > 0 and (u < 32 or rem_s >= m-c):                      # slack constraint
            sw = slack[next]
            ta = 1+below(4); A = ta x (pick(), coef())
            tb = 1+below(2); B = tb x (pickB(), coef())
            j1 = pick(); c1 = rand_fr();  C = [(sw,1),(j1,c1)];  next += 1
        else:                                                            # identity constraint
            x = bits[below(len(bits))]; y = bits[below(len(bits))]; ca = coef(); cb = coef()
            A = [(x,ca),(y,r-ca)]; B = [(x,cb),(y,cb),(0,r-cb)]; C = []
     pick():  s = below(n);                      if cls[s]==SLACK and rank[s] >= next: s = 0
     pickB(): t = below(n); s = t - t%20 + below(7); s = min(s, n-1); same SLACK rule
  witness (stream wseed):  w[0]=1; i in 1..n-1: PUB -> (i==p ? 1951416330 : below(2));
                           BIT -> below(2); SMALL -> below(1024); SLACK -> rand_fr()
     then in constraint order every slack constraint sets w[sw] = a*b - c1*w[j1].
  Duplicate signals inside one row are kept as separate terms (snarkjs sums them at prove time).
"""
from bn254 import R

MASK = (1 << 64) - 1
SEED_NZCP = 0x6E7A6370  # "nzcp"
EXP_EXAMPLE = 1951416330  # 'exp' of the MoH example pass
CONST, PUB, BIT, SMALL, SLACK = range(5)


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


def _coef(rng):
    u = rng.below(100)
    if u < 40:
        return 1
    if u < 80:
        return R - 1
    if u < 95:
        return 1 + rng.below(65535)
    return rng.rand_fr()


def gen_classes(n, p, seed):
    rng = Xoshiro(seed)
    cls = [CONST] + [0] * (n - 1)
    for i in range(1, n):
        if i <= p:
            cls[i] = PUB
        else:
            u = rng.below(100)
            cls[i] = BIT if u < 60 else (SMALL if u < 68 else SLACK)
    return cls


def gen_circuit(n, p, m, seed):
    """Returns (cls, rows, slack_of_row): rows = [(A, B, C)], terms = [(signal, coef)];
    slack_of_row[c] = (sw, j1, c1) for slack constraints, None for identities."""
    cls = gen_classes(n, p, seed)
    rng = Xoshiro(seed + 3)
    slack = [i for i in range(n) if cls[i] == SLACK]
    rank = {s: k for k, s in enumerate(slack)}
    bits = [i for i in range(n) if ((cls[i] == PUB and i < p) or cls[i] == BIT) and i % 20 < 7]
    if not bits:
        bits = [0]
    nxt = 0

    def fix(s):
        return 0 if (cls[s] == SLACK and rank[s] >= nxt) else s

    def pick():
        return fix(rng.below(n))

    def pick_b():
        t = rng.below(n)
        s = t - t % 20 + rng.below(7)
        return fix(min(s, n - 1))

    rows, slack_of_row = [], []
    for c in range(m):
        u = rng.below(100)
        rem = len(slack) - nxt
        if rem > 0 and (u < 32 or rem >= m - c):
            sw = slack[nxt]
            A = []
            for _ in range(1 + rng.below(4)):
                s = pick(); A.append((s, _coef(rng)))
            B = []
            for _ in range(1 + rng.below(2)):
                s = pick_b(); B.append((s, _coef(rng)))
            j1 = pick()
            c1 = rng.rand_fr()
            rows.append((A, B, [(sw, 1), (j1, c1)]))
            slack_of_row.append((sw, j1, c1))
            nxt += 1
        else:
            x = bits[rng.below(len(bits))]
            y = bits[rng.below(len(bits))]
            ca = _coef(rng)
            cb = _coef(rng)
            rows.append(([(x, ca), (y, R - ca)], [(x, cb), (y, cb), (0, R - cb)], []))
            slack_of_row.append(None)
    return cls, rows, slack_of_row


def gen_witness(n, p, cls, rows, slack_of_row, wseed):
    rng = Xoshiro(wseed)
    w = [1] + [0] * (n - 1)
    for i in range(1, n):
        k = cls[i]
        if k == PUB:
            w[i] = EXP_EXAMPLE if i == p else rng.below(2)
        elif k == BIT:
            w[i] = rng.below(2)
        elif k == SMALL:
            w[i] = rng.below(1024)
        else:
            w[i] = rng.rand_fr()
    for (A, B, _), sl in zip(rows, slack_of_row):
        if sl is None:
            continue
        sw, j1, c1 = sl
        a = sum(cf * w[s] for s, cf in A) % R
        b = sum(cf * w[s] for s, cf in B) % R
        w[sw] = (a * b - c1 * w[j1]) % R
    return w


def check_r1cs(rows, w):
    for A, B, C in rows:
        a = sum(cf * w[s] for s, cf in A) % R
        b = sum(cf * w[s] for s, cf in B) % R
        c = sum(cf * w[s] for s, cf in C) % R
        if (a * b - c) % R:
            return False
    return True


def make(n, p, m, seed=SEED_NZCP, wseed=None):
    """Convenience: (rows, witness) with wseed defaulting to seed."""
    cls, rows, sl = gen_circuit(n, p, m, seed)
    w = gen_witness(n, p, cls, rows, sl, seed if wseed is None else wseed)
    return rows, w

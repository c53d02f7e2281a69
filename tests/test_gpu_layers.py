"""GPU parity, layer by layer, through the C ABI: field product -> point addition -> NTT -> MSM.
Each layer is compared bit-exactly with the Python big-int oracle (oracle/bn254.py, groth16.py) on
seeded inputs.  Mirrors the reference's own layering: wasmcurves f1m -> curve -> ffjavascript
fft / multiexp."""
import random

import pytest

import bn254 as b
import formats as f
import groth16 as g

pytestmark = pytest.mark.gpu


def _le(vals):
    return b"".join(f.le(v) for v in vals)


def _ints(buf):
    return [int.from_bytes(buf[i:i + 32], "little") for i in range(0, len(buf), 32)]


@pytest.mark.parametrize("field", [0, 1])
def test_field_ops(amd, field):
    p = b.R if field == 0 else b.Q
    rinv = pow(b.MONT, -1, p)
    rng = random.Random(100 + field)
    n = 4096
    edge = [0, 1, p - 1, p - 2, 2, (1 << 253), b.MONT % p, (p - 1) // 2]
    xs = edge + [rng.randrange(p) for _ in range(n - len(edge))]
    ys = list(reversed(edge)) + [rng.randrange(p) for _ in range(n - len(edge))]
    a, bb = _le(xs), _le(ys)
    assert _ints(amd.field_op(field, 0, a, bb)) == [x * y * rinv % p for x, y in zip(xs, ys)]
    assert _ints(amd.field_op(field, 1, a, bb)) == [(x + y) % p for x, y in zip(xs, ys)]
    assert _ints(amd.field_op(field, 2, a, bb)) == [(x - y) % p for x, y in zip(xs, ys)]
    assert _ints(amd.field_op(field, 3, a, bb)) == [x * b.MONT % p for x in xs]
    assert _ints(amd.field_op(field, 4, a, bb)) == [x * rinv % p for x in xs]
    assert _ints(amd.field_op(field, 5, a, bb)) == [(-x) % p for x in xs]


def _rand_points(curve, cnt, rng):
    gen = b.G1 if curve == 1 else b.G2
    return gen.gen_mul_many([rng.randrange(1, b.R) for _ in range(cnt)])


@pytest.mark.parametrize("curve", [1, 2])
def test_ec_add(amd, curve):
    grp = b.G1 if curve == 1 else b.G2
    enc = f.g1_to_lem if curve == 1 else f.g2_to_lem
    rng = random.Random(7 + curve)
    P = _rand_points(curve, 40, rng)
    Qs = _rand_points(curve, 40, rng)
    # exceptional cases: P+P (doubling), P+(-P) (infinity), inf+Q, P+inf, inf+inf
    P += [P[0], P[1], None, P[2], None]
    Qs += [P[0], grp.neg(P[1]), Qs[0], None, None]
    out = amd.ec_add(curve, b"".join(enc(x) for x in P), b"".join(enc(x) for x in Qs))
    psz = 64 if curve == 1 else 128
    for i, (x, y) in enumerate(zip(P, Qs)):
        exp = grp.add(x, y)
        got = out[i * psz:(i + 1) * psz]
        if exp is None:
            assert got == bytes(psz), i
        elif curve == 1:
            assert got == f.le(exp[0]) + f.le(exp[1]), i
        else:
            assert got == f.le(exp[0][0]) + f.le(exp[0][1]) + f.le(exp[1][0]) + f.le(exp[1][1]), i


@pytest.mark.parametrize("logn", [0, 1, 3, 6, 10, 11, 13])
def test_fft_matches_oracle(amd, logn):
    n = 1 << logn
    rng = random.Random(logn)
    vals = [rng.randrange(b.R) for _ in range(n)]
    mont = _le([v * b.RR % b.R for v in vals])
    rinv = pow(b.RR, -1, b.R)
    fwd = [x * rinv % b.R for x in _ints(amd.fr_fft(mont))]
    assert fwd == g.ntt(vals)
    inv = [x * rinv % b.R for x in _ints(amd.fr_fft(mont, inverse=True))]
    assert inv == g.ntt(vals, inverse=True)


def test_fft_large_roundtrip_and_delta(amd):
    """2^20 (the nzcp_live lower-bound domain): iNTT(NTT(x)) == x, and NTT(delta_1) = w^i."""
    logn = 20
    n = 1 << logn
    rng = random.Random(5)
    raw = rng.randbytes(n * 32)
    # clear the top 3 bits of every element so it is a valid residue (< 2^253 < r)
    arr = bytearray(raw)
    arr[31::32] = bytes(x & 0x1F for x in arr[31::32])
    buf = bytes(arr)
    assert amd.fr_fft(amd.fr_fft(buf), inverse=True) == buf
    delta = bytearray(n * 32)
    delta[32:64] = f.le(b.RR)  # Montgomery 1 at index 1
    out = amd.fr_fft(bytes(delta))
    w = b.fr_root(logn)
    for i in (0, 1, 2, 12345, n - 1):
        assert int.from_bytes(out[i * 32:(i + 1) * 32], "little") == pow(w, i, b.R) * b.RR % b.R


def _msm_case(amd, curve, n, c, rng, special=True):
    grp = b.G1 if curve == 1 else b.G2
    enc = f.g1_to_lem if curve == 1 else f.g2_to_lem
    ks = [rng.randrange(1, b.R) for _ in range(n)]
    bases = grp.gen_mul_many(ks)
    sc = [rng.randrange(b.R) for _ in range(n)]
    if special and n >= 16:
        sc[0] = 0; sc[1] = 1; sc[2] = b.R - 1; sc[3] = 1; sc[4] = 2; sc[5] = (1 << 253) + 5
        sc[6] = (1 << 15); sc[7] = (1 << 16) - 1; sc[8] = 1023; sc[9] = 1
        bases[10] = None; ks[10] = 0               # infinity base
        bases[11] = bases[12]; ks[11] = ks[12]     # repeated base (doubling inside a bucket)
        sc[11] = sc[12]
        bases[13] = grp.neg(bases[14]); ks[13] = (-ks[14]) % b.R
        sc[13] = sc[14]                            # P + (-P) inside one bucket
    out = amd.multiexp(curve, b"".join(enc(x) for x in bases), _le(sc), window_bits=c)
    k = sum(x * y for x, y in zip(ks, sc)) % b.R
    exp = grp.mul(grp.gen, k)
    if exp is None:
        assert out == bytes(len(out))
    elif curve == 1:
        assert out == f.le(exp[0]) + f.le(exp[1])
    else:
        assert out == f.le(exp[0][0]) + f.le(exp[0][1]) + f.le(exp[1][0]) + f.le(exp[1][1])


@pytest.mark.parametrize("n,c", [(1, 0), (3, 4), (64, 5), (300, 8), (1000, 0), (2000, 11), (1500, 16)])
def test_g1_multiexp_known_answer(amd, n, c):
    """MSM KAT: bases [k_i]G => MSM = [sum s_i k_i]G."""
    _msm_case(amd, 1, n, c, random.Random(n * 31 + c))


@pytest.mark.parametrize("n,c", [(2, 0), (100, 6), (700, 0), (600, 13)])
def test_g2_multiexp_known_answer(amd, n, c):
    _msm_case(amd, 2, n, c, random.Random(n * 17 + c))


@pytest.mark.parametrize("curve", [1, 2])
def test_multiexp_exceptional_additions_everywhere(amd, curve):
    """Every bucket addition is a doubling or a cancellation: the same base 150 times with the same scalar,
    then its negative 150 times, then 37 more copies.  The accumulate kernel's fast formula cannot add
    these; its low-limb filter must route every such task to the complete-addition redo pass."""
    grp = b.G1 if curve == 1 else b.G2
    enc = f.g1_to_lem if curve == 1 else f.g2_to_lem
    rng = random.Random(4242 + curve)
    k = rng.randrange(1, b.R)
    P = grp.mul(grp.gen, k)
    s0 = rng.randrange(b.R)
    bases = [P] * 150 + [grp.neg(P)] * 150 + [P] * 37
    ks = [k] * 150 + [(-k) % b.R] * 150 + [k] * 37
    for c in (0, 7, 16):
        out = amd.multiexp(curve, b"".join(enc(x) for x in bases), _le([s0] * len(bases)), window_bits=c)
        exp = grp.mul(grp.gen, sum(x * s0 for x in ks) % b.R)
        want = (f.le(exp[0]) + f.le(exp[1])) if curve == 1 else b"".join(f.le(v) for v in (exp[0][0], exp[0][1], exp[1][0], exp[1][1]))
        assert out == want


def test_multiexp_nzcp_like_scalar_mix(amd):
    """60 % bits / 8 % small / 32 % full-width: one huge bucket + task splitting."""
    rng = random.Random(99)
    n = 3000
    ks = [rng.randrange(1, b.R) for _ in range(n)]
    bases = b.G1.gen_mul_many(ks)
    sc = []
    for _ in range(n):
        u = rng.randrange(100)
        sc.append(0 if u < 30 else 1 if u < 60 else rng.randrange(1024) if u < 68 else rng.randrange(b.R))
    out = amd.multiexp(1, b"".join(f.g1_to_lem(x) for x in bases), _le(sc), window_bits=0)
    exp = b.G1.mul(b.G1_GEN, sum(x * y for x, y in zip(ks, sc)) % b.R)
    assert out == f.le(exp[0]) + f.le(exp[1])

"""GPU layer tests of the arithmetic the hot path actually runs ("What's weak" 2).

The MSM / NTT / QAP kernels compute on the 9 x 29-bit lazy field (csrc/fq29.cuh, fr29.cuh) and the XYZZ
formulas over it (csrc/ec29.cuh); the canonical 8 x 32 field of test_gpu_layers.py::test_field_ops only
serves the host tail.  Here every lazy-field primitive and every point formula runs ON THE DEVICE through
the C ABI (g16_f29_op / g16_x29_op / g16_qap_eval) and is compared with the Python big-int oracle
(oracle/bn254.py, oracle/groth16.py), including inputs at the bounds the kernels rely on (products of
values below 16p; accumulator X below 5.4p, Y below 3.6p, affine coordinates below 2p) and the
exceptional additions the hot loop flags for its redo pass.

These are the device twins of wasmcurves 0.1.0 f1m_mul / f1m_square / f1m_add / f1m_sub, the curve add /
double of build_curve_jacobian_a0.js and snarkjs buildABC1 (pins /root/reference/yarn.lock:1132-1138,
987-1001); exact for integer work: congruence mod p plus the documented value bound of every result."""
import random

import pytest

import bn254 as b
import formats as f
import groth16 as g

pytestmark = pytest.mark.gpu

RAD = 1 << 261


def _mod(field):
    return b.R if field == 0 else b.Q


def _lazy(rng, p, kmax):
    """a residue in lazy form: m + k p with k <= kmax (value below (kmax + 1) p)."""
    return rng.randrange(p) + rng.randrange(kmax + 1) * p


@pytest.mark.parametrize("field", [0, 1])
def test_f29_products(amd, field):
    p = _mod(field)
    rinv = pow(RAD, -1, p)
    rng = random.Random(2900 + field)
    n = 2048
    edge = [0, 1, p - 1, p, p + 1, 16 * p - 1, 15 * p + 12345, (1 << 232) - 1, (1 << 29) - 1, 8 * p]
    xs = edge + [_lazy(rng, p, 15) for _ in range(n - len(edge))]
    ys = list(reversed(edge)) + [_lazy(rng, p, 15) for _ in range(n - len(edge))]
    for op in (0, 9):   # product-scanning and row-wise Montgomery products
        out = amd.f29_op(field, op, xs, ys)
        for x, y, o in zip(xs, ys, out):
            assert o % p == x * y * rinv % p
            assert o <= x * y // RAD + p          # documented bound: a b / 2^261 + p  (< 2.51 p for inputs < 16 p)
            assert o < 251 * p // 100
    out = amd.f29_op(field, 1, xs)
    for x, o in zip(xs, out):
        assert o % p == x * x * rinv % p and o <= x * x // RAD + p
    # fused forms: a b + c d and a^2 + c d under ONE reduction (Fq2 products, the Y3 of the hot-loop addition);
    # their call sites keep the operands below 8 p
    a8 = [v % (8 * p) for v in xs]
    b8 = [v % (8 * p) for v in ys]
    cs = [_lazy(rng, p, 7) for _ in range(n)]
    ds = [_lazy(rng, p, 7) for _ in range(n)]
    out = amd.f29_op(field, 2, a8, b8, cs, ds)
    for x, y, c, d, o in zip(a8, b8, cs, ds, out):
        assert o % p == (x * y + c * d) * rinv % p and o <= (x * y + c * d) // RAD + p
    out = amd.f29_op(field, 3, a8, None, cs, ds)
    for x, c, d, o in zip(a8, cs, ds, out):
        assert o % p == (x * x + c * d) * rinv % p and o <= (x * x + c * d) // RAD + p


@pytest.mark.parametrize("field", [0, 1])
def test_f29_additive_ops_and_zero_tests(amd, field):
    p = _mod(field)
    rng = random.Random(2950 + field)
    n = 1024
    xs = [0, p, 13 * p + 5] + [_lazy(rng, p, 12) for _ in range(n - 3)]
    ys = [0, 2 * p, p - 1] + [_lazy(rng, p, 1) for _ in range(n - 3)]        # subtrahends up to 2 p
    assert amd.f29_op(field, 4, xs, ys) == [x + y for x, y in zip(xs, ys)]    # limb-wise add + carry ripple: exact
    assert amd.f29_op(field, 5, xs, ys) == [x + 2 * p - y for x, y in zip(xs, ys)]
    y6 = [6 * p, 0] + [_lazy(rng, p, 5) for _ in range(n - 2)]                # subtrahends up to 6 p
    assert amd.f29_op(field, 6, xs, y6) == [x + 6 * p - y for x, y in zip(xs, y6)]
    # X3 = R^2 + 4p - PPP - 2Q in one pass: needs PPP + 2 Q <= 4 p
    bs = [_lazy(rng, p, 0) for _ in range(n)]
    cs = [rng.randrange(3 * p // 2) for _ in range(n)]
    bs[0], cs[0] = p, 3 * p // 2
    assert amd.f29_op(field, 7, xs, bs, cs) == [x + 4 * p - y - 2 * c for x, y, c in zip(xs, bs, cs)]
    assert amd.f29_op(field, 8, ys) == [2 * p - y for y in ys]
    # zero tests: exact congruence test and the hot loop's low-limb filter (must fire on every k p, k <= 7)
    zs = [k * p for k in range(16)] + [k * p + 1 for k in range(16)] + [_lazy(rng, p, 15) for _ in range(64)]
    out = amd.f29_op(field, 11, zs)
    for z, o in zip(zs, out):
        assert (o & 1) == (1 if z % p == 0 else 0)
        if z % p == 0 and z <= 7 * p:
            assert (o >> 29) & 1 == 1
    if field == 0:
        # NTT / QAP weak reduction: any value below 16 r comes back below 1.0001 r, same residue
        ws = [0, p - 1, p, 16 * p - 1] + [_lazy(rng, p, 15) for _ in range(n)]
        out = amd.f29_op(0, 10, ws)
        for w, o in zip(ws, out):
            assert o % p == w % p and o < p + (p >> 13)


# ---------------------------------------------------------------------------------------------------------
def _grp(curve):
    return b.G1 if curve == 1 else b.G2


def _to_m(curve, v):
    """field element -> Montgomery(2^261) residue(s)"""
    if curve == 1:
        return v * RAD % b.Q
    return (v[0] * RAD % b.Q, v[1] * RAD % b.Q)


def _lift(curve, v, k, rng):
    """add up to k multiples of p to every component (lazy representation at its bound)"""
    if curve == 1:
        return v + rng.randrange(k + 1) * b.Q
    return (v[0] + rng.randrange(k + 1) * b.Q, v[1] + rng.randrange(k + 1) * b.Q)


def _xyzz_of(curve, P, rng, kx=4, ky=2):
    """a random XYZZ representation of affine P with X below (kx + 1) p, Y below (ky + 1) p"""
    F = _grp(curve).F
    if P is None:
        z = F.zero
        return (z, z, z, z) if curve == 1 else ((0, 0),) * 4
    zr = rng.randrange(1, b.Q) if curve == 1 else (rng.randrange(1, b.Q), rng.randrange(b.Q))
    zz = F.mul(zr, zr)
    zzz = F.mul(zz, zr)
    return (_lift(curve, _to_m(curve, F.mul(P[0], zz)), kx, rng), _lift(curve, _to_m(curve, F.mul(P[1], zzz)), ky, rng),
            _to_m(curve, zz), _to_m(curve, zzz))


def _affine_m(curve, P, rng, k=1):
    return (_lift(curve, _to_m(curve, P[0]), k, rng), _lift(curve, _to_m(curve, P[1]), k, rng))


def _decode(curve, pt):
    """XYZZ (Montgomery-261 lazy) -> affine point or None"""
    F = _grp(curve).F
    rinv = pow(RAD, -1, b.Q)
    if curve == 1:
        x, y, zz, zzz = [v * rinv % b.Q for v in pt]
        if pt[2] == 0:
            return None
    else:
        x, y, zz, zzz = [(v[0] * rinv % b.Q, v[1] * rinv % b.Q) for v in pt]
        if pt[2] == (0, 0):
            return None
    assert F.mul(F.mul(zz, zz), zz) == F.mul(zzz, zzz), "ZZ^3 != ZZZ^2"
    return (F.mul(x, F.inv(zz)), F.mul(y, F.inv(zzz)))


def _check_bounds(curve, pt):
    """documented coordinate bounds of the accumulator (ec29.cuh header)"""
    comps = [(c,) if curve == 1 else c for c in pt]
    assert all(v < 54 * b.Q // 10 for v in comps[0])
    assert all(v < 36 * b.Q // 10 for v in comps[1])
    assert all(v < 251 * b.Q // 100 for v in comps[2] + comps[3])


@pytest.mark.parametrize("curve", [1, 2])
def test_x29_madd_fast_and_complete(amd, curve):
    grp = _grp(curve)
    rng = random.Random(2960 + curve)
    n = 96
    Ps = grp.gen_mul_many([rng.randrange(1, b.R) for _ in range(n)])
    Qs = grp.gen_mul_many([rng.randrange(1, b.R) for _ in range(n)])
    # exceptional pairs at the end: doubling, cancellation
    Ps += [Ps[0], Ps[1], Ps[2], Ps[3]]
    Qs += [Ps[0], grp.neg(Ps[1]), Ps[2], grp.neg(Ps[3])]
    # G1's hot-loop formula folds -Y1 into a fused product and needs Y1 <= 2p (an affine coordinate or an earlier
    # fused Y3 < 1.3p: ec29.cuh x29_madd_fast); G2 subtracts with K = 4
    acc = [_xyzz_of(curve, P, rng, ky=1 if curve == 1 else 2) for P in Ps]
    q = [_affine_m(curve, Q, rng) for Q in Qs]
    out, flags = amd.x29_op(curve, 0, acc, q)           # the hot-loop formula
    for i in range(n):
        assert _decode(curve, out[i]) == grp.add(Ps[i], Qs[i]), i
        _check_bounds(curve, out[i])
        if curve == 1:
            assert out[i][1] < 13 * b.Q // 10         # the fused Y3 bound the next iteration relies on
    assert sum(flags[:n]) <= 1                            # the low-limb filter fires on ~2^-26 of ordinary additions
    assert flags[n:] == [1, 1, 1, 1]                      # doubling / cancellation MUST be sent to the redo pass
    out, _ = amd.x29_op(curve, 1, acc, q)                # complete mixed addition (msm_redo_kernel)
    for i, (P, Q) in enumerate(zip(Ps, Qs)):
        assert _decode(curve, out[i]) == grp.add(P, Q), i
    out, _ = amd.x29_op(curve, 4, acc, q)                # through the packed 64 / 128-byte resident base format
    for i, (P, Q) in enumerate(zip(Ps, Qs)):
        assert _decode(curve, out[i]) == grp.add(P, Q), i
    # chained additions: feed the (lazy, unreduced) output back in -- bounds must hold along a task
    cur, pts = acc[:n], Ps[:n]
    for step in range(6):
        nxt = grp.gen_mul_many([rng.randrange(1, b.R) for _ in range(n)])
        cur, flags = amd.x29_op(curve, 0, cur, [_affine_m(curve, Q, rng) for Q in nxt])
        pts = [grp.add(P, Q) for P, Q in zip(pts, nxt)]
        assert sum(flags) <= 1
        for i in range(n):
            assert _decode(curve, cur[i]) == pts[i], (step, i)
            _check_bounds(curve, cur[i])
    # infinity accumulator: the complete formula starts from q
    zero = _xyzz_of(curve, None, rng)
    out, _ = amd.x29_op(curve, 1, [zero] * 4, q[:4])
    assert [_decode(curve, o) for o in out] == Qs[:4]


@pytest.mark.parametrize("curve", [1, 2])
def test_x29_add_and_dbl(amd, curve):
    grp = _grp(curve)
    rng = random.Random(2970 + curve)
    n = 64
    Ps = grp.gen_mul_many([rng.randrange(1, b.R) for _ in range(n)])
    Qs = grp.gen_mul_many([rng.randrange(1, b.R) for _ in range(n)])
    Ps += [Ps[0], Ps[1], None, Ps[2], None]
    Qs += [Ps[0], grp.neg(Ps[1]), Qs[0], None, None]
    acc = [_xyzz_of(curve, P, rng) for P in Ps]
    q = [_xyzz_of(curve, Q, rng) for Q in Qs]
    out, _ = amd.x29_op(curve, 2, acc, q)                # add-2008-s incl. doubling, cancellation, infinities
    for i, (P, Q) in enumerate(zip(Ps, Qs)):
        assert _decode(curve, out[i]) == grp.add(P, Q), i
    out, _ = amd.x29_op(curve, 3, acc)                   # dbl-2008-s-1
    for i, P in enumerate(Ps):
        assert _decode(curve, out[i]) == grp.add(P, P), i
    # reduction-tree shape: sums of sums (what msm_bucket_reduce / wave_reduce do with unreduced outputs)
    cur, pts = acc[:n], Ps[:n]
    for step in range(5):
        half = len(cur) // 2
        cur, _ = amd.x29_op(curve, 2, cur[:half], cur[half:2 * half])
        pts = [grp.add(x, y) for x, y in zip(pts[:half], pts[half:2 * half])]
        for i in range(half):
            assert _decode(curve, cur[i]) == pts[i], (step, i)


# ---------------------------------------------------------------------------------------------------------
def _qap_check(amd, zkey, wtns):
    zk = f.read_zkey(zkey)
    w = f.read_wtns(wtns)["w"]
    prover = amd.Prover(zkey)
    prover.stage(0, wtns)
    got = prover.qap_eval(0)
    prover.close()
    rinv = pow(b.MONT, -1, b.R)
    exp = g.build_abc(zk, w)
    for name, gv, ev in zip("ABC", got, exp):
        assert [x * rinv % b.R for x in gv] == ev, f"{name}_T differs from buildABC1"
    return zk


def test_qap_eval_short_rows(amd):
    """buildABC1 operator on the synthetic NZCP-shaped circuit (rows of <= 4 terms: one lane per row)."""
    zkey, wtns, _ = amd.synth_setup(700, 13, 600, 4711)
    _qap_check(amd, zkey, wtns)


def test_qap_eval_long_rows(amd):
    """... and on a real SHA-256 circuit, whose modular-addition rows carry ~260 terms (one wavefront per row)."""
    out = amd.sha256_message_setup(b"nzcp", 17)
    zk = _qap_check(amd, out["zkey"], out["wtns"])
    per_row = {}
    for (m, c, s, v) in zk["coefs"]:
        per_row[(m, c)] = per_row.get((m, c), 0) + 1
    assert max(per_row.values()) > 64      # the wavefront path was really exercised


def test_qap_eval_plus_minus_one_records(amd):
    """Records whose coefficient is +1 or -1 (79 % of the real circuit's) add or subtract the Montgomery image of their witness
    word instead of multiplying (qap_row_sum): rows of only +1, only -1 (long runs of subtractions: the lazy sums' bound), mixed
    with general coefficients, in all three row shapes (a lane, eight lanes, a wavefront per row), witness words up to r - 1."""
    import random
    rnd = random.Random(77)
    n, p = 600, 3
    coefs = [1, b.R - 1, 1, b.R - 1, 2, b.R - 2, rnd.randrange(b.R), 1]

    def lin(k, pick):
        return [(rnd.randrange(n), pick()) for _ in range(k)]
    rows = []
    for k in (1, 2, 3, 9, 16):                       # a lane per row
        rows.append((lin(k, lambda: 1), lin(k, lambda: b.R - 1), []))
        rows.append((lin(k, lambda: b.R - 1), lin(k, lambda: rnd.choice(coefs)), []))
    for k in (17, 23, 40, 64):                       # eight lanes per row
        rows.append((lin(k, lambda: b.R - 1), lin(2, lambda: 1), []))
        rows.append((lin(k, lambda: rnd.choice(coefs)), lin(k, lambda: rnd.choice(coefs)), []))
    for k in (65, 130, 300):                         # a wavefront per row
        rows.append((lin(k, lambda: b.R - 1), lin(k, lambda: 1), []))
        rows.append((lin(k, lambda: rnd.choice(coefs)), lin(1, lambda: rnd.randrange(b.R)), []))
    rows += [([], lin(3, lambda: 1), []), (lin(3, lambda: b.R - 1), [], [])]
    zk, _ = g.setup(n, p, rows, g.trapdoor(78))
    w = [1] + [rnd.choice([b.R - 1, b.R - 2, rnd.randrange(b.R), rnd.randrange(256), 0, 1]) for _ in range(n - 1)]
    zkb, wt = f.write_zkey(zk), f.write_wtns(w)
    _qap_check(amd, zkb, wt)
    # ... and the whole proof on it (the witness satisfies nothing: parity of the arithmetic)
    prover = amd.Prover(zkb)
    proof, pub = prover.prove(wt, f.le(5), f.le(6))
    prover.close()
    (A, B, C), opub = g.prove(zk, w, 5, 6)
    assert proof == f.proof_obj(A, B, C) and pub == [str(x) for x in opub]

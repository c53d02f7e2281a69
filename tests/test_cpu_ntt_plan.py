"""Python model of ntt.hip's pass plan and tile index math (the part of the NTT that cannot be
debugged without a GPU): DIT passes over bit-reversed input must equal the oracle NTT, DIF passes in
reverse order must equal the oracle inverse NTT up to bit reversal and 1/N.  Mirrors
ntt_tables_create (plan) and ntt_pass_kernel (gidx / butterfly / twiddle index)."""
import random

import pytest

from bn254 import R, fr_root
from groth16 import ntt


def plan(L, tile_log_max=10, min_tb=2):
    tl = min(tile_log_max, L)
    passes = [(0, tl, 0)]
    done = tl
    while done < L:
        S = min(L - done, tl - min_tb)
        tb = tl - S
        passes.append((done, S, tb))
        done += S
    return tl, passes


def run_pass(x, tw, L, tl, lo_bits, S, tb, dif):
    N, tile, T = 1 << L, 1 << tl, 1 << tb
    for t in range(N // tile):
        if lo_bits == 0:
            base = t * tile
        else:
            per = (1 << lo_bits) >> tb
            base = ((t // per) << (lo_bits + S)) + (t % per) * T
        gidx = lambda e: base + ((e >> tb) << lo_bits) + (e & (T - 1))  # noqa: E731
        lds = [x[gidx(e)] for e in range(tile)]
        for k in range(S):
            st = S - 1 - k if dif else k
            bit, beta = tb + st, lo_bits + st
            for bf in range(tile // 2):
                e0 = ((bf >> bit) << (bit + 1)) | (bf & ((1 << bit) - 1))
                e1 = e0 | (1 << bit)
                w = tw[(gidx(e0) & ((1 << beta) - 1)) << (L - 1 - beta)]
                u, v = lds[e0], lds[e1]
                if dif:
                    lds[e0], lds[e1] = (u + v) % R, (u - v) * w % R
                else:
                    v = v * w % R
                    lds[e0], lds[e1] = (u + v) % R, (u - v) % R
        for e in range(tile):
            x[gidx(e)] = lds[e]


def bitrev(i, L):
    return int(bin(i)[2:].zfill(L)[::-1], 2) if L else 0


@pytest.mark.parametrize("L,tlmax", [(1, 10), (3, 10), (5, 4), (7, 4), (9, 5), (11, 10)])
def test_pass_plan_matches_oracle(L, tlmax):
    rng = random.Random(L)
    N = 1 << L
    w = fr_root(L)
    tw = [pow(w, i, R) for i in range(max(1, N // 2))]
    twi = [pow(w, -i, R) for i in range(max(1, N // 2))]
    v = [rng.randrange(R) for _ in range(N)]
    tl, passes = plan(L, tlmax)
    x = [v[bitrev(i, L)] for i in range(N)]
    for lo, S, tb in passes:
        run_pass(x, tw, L, tl, lo, S, tb, False)
    assert x == ntt(v)
    y = list(v)
    for lo, S, tb in reversed(passes):
        run_pass(y, twi, L, tl, lo, S, tb, True)
    ninv = pow(N, -1, R)
    assert [y[bitrev(i, L)] * ninv % R for i in range(N)] == ntt(v, inverse=True)

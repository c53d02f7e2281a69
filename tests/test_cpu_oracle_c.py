"""Pins the oracle's C restatement (oracle/c/g16_oracle.c -- the CPU baseline of bench.py and the
big-case checker) against the Python big-int oracle, the golden fixtures and the trapdoor KAT."""
import ctypes
import json
import os
import subprocess

import pytest

import bn254 as b
import formats as f
import groth16 as g
import synth
from conftest import ROOT, golden_path


@pytest.fixture(scope="module")
def olib():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle", "c")], stdout=subprocess.DEVNULL)
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_build", "libg16oracle.so"))
    lib.g16o_prove.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                               ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
    return lib


def c_prove(olib, zk, wt, r, s, p, threads=4):
    out = ctypes.create_string_buffer(256)
    pub = ctypes.create_string_buffer(max(1, p * 32))
    assert olib.g16o_prove(zk, len(zk), wt, len(wt), f.le(r), f.le(s), out, pub, threads) == 0
    o = out.raw
    v = [int.from_bytes(o[i * 32:(i + 1) * 32], "little") for i in range(8)]
    proof = ((v[0], v[1]), ((v[2], v[3]), (v[4], v[5])), (v[6], v[7]))
    return proof, [int.from_bytes(pub.raw[i * 32:(i + 1) * 32], "little") for i in range(p)]


@pytest.mark.parametrize("name", ["tiny", "small", "nzcp513"])
def test_c_oracle_reproduces_golden(olib, name):
    zk = open(golden_path(name + ".zkey"), "rb").read()
    wt = open(golden_path(name + ".wtns"), "rb").read()
    meta = json.load(open(golden_path(name + ".json")))
    proof, pub = c_prove(olib, zk, wt, int(meta["r"]), int(meta["s"]), meta["p"])
    assert f.proof_obj(*proof) == meta["proof"]
    assert [str(x) for x in pub] == meta["public"]


@pytest.mark.parametrize("n,p,m,seed,threads", [(50, 3, 33, 41, 1), (400, 10, 350, 42, 3)])
def test_c_oracle_equals_python_oracle(olib, n, p, m, seed, threads):
    rows, w = synth.make(n, p, m, seed)
    zk, sec = g.setup(n, p, rows, g.trapdoor(seed + 1))
    rng = synth.Xoshiro(seed + 2)
    r, s = rng.rand_fr(), rng.rand_fr()
    proof, pub = c_prove(olib, f.write_zkey(zk), f.write_wtns(w), r, s, p, threads)
    exp, epub = g.prove(zk, w, r, s)
    assert proof == exp and pub == epub
    assert proof == g.expected_proof(sec, p, w, r, s)


def test_c_oracle_trapdoor_kat_2_14(amd, olib):
    """Beyond the big-int prover's reach: 2^14 domain, pinned by the scalar-only trapdoor KAT."""
    n, p, m, seed = 14000, 513, 14000, 55
    zkey, wtns, _ = amd.synth_setup(n, p, m, seed, 0)
    rows, w = synth.make(n, p, m, seed)
    td = g.trapdoor(seed + 1)
    L = g.lagrange_at(1 << 14, td["tau"])
    u = [0] * n; v = [0] * n; t = [0] * n
    for c, (A, B, C) in enumerate(rows):
        for sg, cf in A: u[sg] = (u[sg] + cf * L[c]) % b.R
        for sg, cf in B: v[sg] = (v[sg] + cf * L[c]) % b.R
        for sg, cf in C: t[sg] = (t[sg] + cf * L[c]) % b.R
    for i in range(p + 1):
        u[i] = (u[i] + L[m + i]) % b.R
    rng = synth.Xoshiro(seed + 2)
    r, s = rng.rand_fr(), rng.rand_fr()
    proof, pub = c_prove(olib, zkey, wtns, r, s, p, 8)
    assert proof == g.expected_proof({"u": u, "v": v, "t": t, **td}, p, w, r, s)
    assert pub == w[1:p + 1]

"""CPU suite for the product's host side: library loads and exports every declared symbol, the
host-compiled product arithmetic (fp.cuh/ec.cuh) reproduces the oracle via the setup tool, the
snarkjs error texts come out before any GPU is touched, and compute entry points refuse to run
without a HIP device (no CPU fallback)."""
import json
import os
import re

import pytest

import formats as f
import groth16 as g
import synth
from conftest import ROOT, golden_path


def test_library_exports_every_declared_symbol(amd):
    hdr = open(os.path.join(ROOT, "include", "g16_prover.h")).read()
    declared = set(re.findall(r"\b(g16_[a-z0-9_]+)\s*\(", hdr))
    declared -= {"g16_opts", "g16_proof", "g16_info", "g16_timings"}
    assert declared == set(amd.EXPORTS), declared ^ set(amd.EXPORTS)
    lib = amd.load()
    for name in declared:
        assert getattr(lib, name) is not None


@pytest.mark.parametrize("n,p,m,seed", [(24, 2, 12, 1), (150, 6, 120, 2), (333, 20, 300, 9)])
def test_setup_tool_equals_python_oracle(amd, n, p, m, seed):
    """Same seed => byte-identical zkey/wtns from the C++ tool (product fp/ec arithmetic on the
    host, fixed-base tables, batch inversion) and from oracle/groth16.py + oracle/formats.py."""
    zkey, wtns, vkey = amd.synth_setup(n, p, m, seed, 4)
    rows, w = synth.make(n, p, m, seed)
    zk, _ = g.setup(n, p, rows, g.trapdoor(seed + 1))
    assert wtns == f.write_wtns(w)
    assert zkey == f.write_zkey(zk)
    exp_vk = f.g1_to_lem(zk["alpha1"]) + f.g2_to_lem(zk["beta2"]) + f.g2_to_lem(zk["gamma2"]) + \
        f.g2_to_lem(zk["delta2"]) + b"".join(f.g1_to_lem(P) for P in zk["IC"])
    assert vkey == exp_vk
    w2 = amd.synth_witness(n, p, m, seed, 4242)
    assert w2 == f.write_wtns(synth.make(n, p, m, seed, 4242)[1])


def test_setup_tool_reproduces_golden(amd):
    meta = json.load(open(golden_path("small.json")))
    zkey, wtns, _ = amd.synth_setup(meta["n"], meta["p"], meta["m"], meta["seed"], 2)
    assert zkey == open(golden_path("small.zkey"), "rb").read()
    assert wtns == open(golden_path("small.wtns"), "rb").read()


def test_setup_tool_wide_table_path(amd):
    """n >= 20000 switches the fixed-base tables to 16-bit windows; spot-check points vs the oracle."""
    import bn254 as b
    n, p, m, seed = 20000, 3, 20, 31
    zkey, _, _ = amd.synth_setup(n, p, m, seed, 8)
    rows, _ = synth.make(n, p, m, seed)
    td = g.trapdoor(seed + 1)
    L = g.lagrange_at(32, td["tau"])
    secs = f.read_binfile(zkey, "zkey", 2)
    A = f.section(zkey, secs, 5)
    u = {}
    for c, (Ar, _, _) in enumerate(rows):
        for s, cf in Ar:
            u[s] = (u.get(s, 0) + cf * L[c]) % b.R
    for i in range(p + 1):
        u[i] = (u.get(i, 0) + L[m + i]) % b.R
    for s in list(u)[:12]:
        assert f.g1_from_lem(A[s * 64:(s + 1) * 64]) == b.G1.mul(b.G1_GEN, u[s])
    assert A[64 * 19999:] == bytes(64) or 19999 in u


def test_create_errors_match_snarkjs_without_gpu(amd):
    zk = open(golden_path("tiny.zkey"), "rb").read()
    with pytest.raises(amd.G16Error, match="zkey: Invalid File format"):
        amd.Prover(b"wtns" + zk[4:])
    bad = bytearray(zk); bad[4] = 9
    with pytest.raises(amd.G16Error, match="Version not supported"):
        amd.Prover(bytes(bad))
    secs = f.read_binfile(zk, "zkey", 2)
    bad = bytearray(zk); bad[secs[1][0][0]] = 2
    with pytest.raises(amd.G16Error, match="zkey file is not groth16"):
        amd.Prover(bytes(bad))
    bad = bytearray(zk); bad[secs[2][0][0] + 4] ^= 1          # q of another curve
    with pytest.raises(amd.G16Error, match="Curve not supported"):
        amd.Prover(bytes(bad))
    with pytest.raises(amd.G16Error, match="zkey: Invalid File format"):
        amd.Prover(zk[:200])


def test_no_cpu_fallback(amd):
    """On a box without a HIP device every compute entry point must fail loudly."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the -m gpu tests")
    zk = open(golden_path("tiny.zkey"), "rb").read()
    with pytest.raises(amd.G16Error) as e:
        amd.Prover(zk)
    assert e.value.code == -4
    with pytest.raises(amd.G16Error) as e:
        amd.fr_fft(bytes(64))
    assert e.value.code == -4
    with pytest.raises(amd.G16Error) as e:
        amd.multiexp(1, bytes(64), bytes(32))
    assert e.value.code == -4
    # verifier, PLONK prover and PLONK setup: no CPU path either
    zkey, _, vkey = amd.synth_setup(150, 6, 120, 2, 2)
    with pytest.raises(amd.G16Error) as e:
        amd.Verifier(vkey, 6)
    assert e.value.code == -4
    import plonk as pk
    rows, _w = synth.make(24, 2, 12, 1)
    pz = pk.write_zkey(pk.setup(24, 2, rows, tau=5))
    with pytest.raises(amd.G16Error) as e:
        amd.PlonkProver(pz)
    assert e.value.code == -4
    with pytest.raises(amd.G16Error) as e:
        amd.plonk_setup(f.write_r1cs(24, 2, 0, synth.gen_circuit(24, 2, 12, 1)[1]), 1, device=0)
    assert e.value.code == -4
    # the device path of the trapdoor setup, once selected, does not drop back to the host threads either
    amd.setup_device(0)
    try:
        with pytest.raises(amd.G16Error) as e:
            amd.synth_setup(150, 6, 120, 2, 2)
        assert e.value.code == -4
    finally:
        amd.setup_device(-1)
    assert len(amd.synth_setup(150, 6, 120, 2, 2)[0]) > 0


@pytest.mark.parametrize("n,p,m,seed", [(24, 2, 12, 1), (200, 7, 160, 13)])
def test_r1cs_setup_equals_synth_setup(amd, n, p, m, seed):
    """the .r1cs reader + trapdoor setup.  The synthetic circuit written as an iden3
    .r1cs by the oracle and set up by g16_r1cs_setup must give the byte-identical zkey that
    g16_synth_setup (and the Python oracle setup) produce for the same seed."""
    _, rows, _ = synth.gen_circuit(n, p, m, seed)
    r1cs = f.write_r1cs(n, p, 0, rows)
    zkey, vkey = amd.r1cs_setup(r1cs, seed, 2)
    zkey2, _, vkey2 = amd.synth_setup(n, p, m, seed, 2)
    assert zkey == zkey2 and vkey == vkey2
    # public inputs split between outputs and inputs gives the same nPublic
    zkey3, _ = amd.r1cs_setup(f.write_r1cs(n, 1, p - 1, rows), seed, 1)
    assert zkey3 == zkey
    # the snarkjs-shaped verification key of the setup equals the oracle's
    rows_w = synth.make(n, p, m, seed)[0]
    zk, _ = g.setup(n, p, rows_w, g.trapdoor(seed + 1))
    assert amd.vkey_json(vkey, p) == f.vkey_obj(zk)


def test_r1cs_reader_errors(amd):
    with pytest.raises(amd.G16Error, match="r1cs: Invalid File format"):
        amd.r1cs_setup(b"zkey" + bytes(40), 1)
    _, rows, _ = synth.gen_circuit(24, 2, 12, 1)
    good = f.write_r1cs(24, 2, 0, rows)
    with pytest.raises(amd.G16Error, match="truncated"):
        amd.r1cs_setup(good[:len(good) // 2], 1)
    bad = bytearray(good)
    secs = f.read_binfile(good, "r1cs", 1)
    bad[secs[1][0][0] + 4] ^= 1      # another prime
    with pytest.raises(amd.G16Error, match="not the bn128 scalar field"):
        amd.r1cs_setup(bytes(bad), 1)


def test_nzcp_fixed_circuit_tool(tmp_path):
    """tools/nzcp_fixed_circuit.py (host only): artefacts for snarkjs cross-checks from the example pass URI."""
    import subprocess
    import sys
    from conftest import ROOT, golden_path
    out = tmp_path / "fx"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "nzcp_fixed_circuit.py"),
                        golden_path("example_pass_uri.txt"), str(out)], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr
    assert "Jack,Sparrow,1960-04-16" in r.stdout and "exp 1951416330" in r.stdout
    exp = json.load(open(out / "expected_public.json"))
    assert len(exp) == 513 and exp[512] == "1951416330"
    assert "".join(exp[256:512]) == bin(int("271ce33d671a2d3b816d788135f4343e14bc66802f8cd841faac939e8c11f3ee", 16))[2:].zfill(256)
    vk = json.load(open(out / "verification_key.json"))
    assert vk["nPublic"] == 513 and len(vk["IC"]) == 514 and vk["protocol"] == "groth16"
    assert (out / "circuit.r1cs").read_bytes()[:4] == b"r1cs" and (out / "circuit.zkey").read_bytes()[:4] == b"zkey"


def test_parsers_survive_mutated_keys(amd):
    """The C ABI takes untrusted buffers: thousands of mutated Groth16 and PLONK keys (flipped header bytes, truncations,
    overwritten 32-bit fields, absurd section sizes) must come back as G16_E_FORMAT -- or pass the parser and stop at
    the device check on this GPU-less box -- never crash."""
    import random
    import struct
    import torch
    import plonk as pk
    if torch.cuda.is_available():
        pytest.skip("GPU present: a mutated key that parses would go on to the device")
    z = open(golden_path("tiny.zkey"), "rb").read()
    rows, _w = synth.make(24, 2, 12, 1)
    pz = pk.write_zkey(pk.setup(24, 2, rows, tau=5))
    rng = random.Random(1)
    for buf, ctor in ((z, amd.Prover), (pz, amd.PlonkProver)):
        codes = set()
        for _ in range(1500):
            b = bytearray(buf)
            k = rng.randrange(4)
            if k == 0:
                for _j in range(rng.randrange(1, 4)):
                    b[rng.randrange(min(len(b), 600))] = rng.randrange(256)
            elif k == 1:
                b = b[:rng.randrange(len(b))]
            elif k == 2:
                i = rng.randrange(len(b) - 4)
                b[i:i + 4] = struct.pack("<I", rng.choice([0, 1, 0xffffffff, 0x7fffffff, rng.randrange(1 << 32)]))
            else:
                i = 12 + rng.randrange(200)
                b[i:i + 8] = struct.pack("<Q", rng.choice([0, 1, len(b), 1 << 40, (1 << 64) - 1]))
            with pytest.raises(amd.G16Error) as e:
                ctor(bytes(b))
            codes.add(e.value.code)
        assert codes <= {-2, -4, -1} and -2 in codes


def test_r1cs_and_ptau_readers_survive_mutated_files(amd):
    """Mutated .r1cs / .ptau images: an error (or a valid setup), never a crash or an allocation sized by an untrusted
    header field (r02: a mutated wire count made the reader allocate 100+ GB)."""
    import random
    import struct
    import torch
    import plonk as pk
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    _, rows, _ = synth.gen_circuit(60, 5, 40, 3)
    r = f.write_r1cs(60, 5, 0, rows)
    pt = pk.write_ptau(6, 777)
    rng = random.Random(2)

    def mutate(buf, head):
        b = bytearray(buf)
        k = rng.randrange(4)
        if k == 0:
            for _j in range(rng.randrange(1, 4)):
                b[rng.randrange(min(len(b), head))] = rng.randrange(256)
        elif k == 1:
            b = b[:rng.randrange(len(b))]
        elif k == 2:
            i = rng.randrange(min(len(b) - 4, head))
            b[i:i + 4] = struct.pack("<I", rng.choice([0, 1, 0xffffffff, 0x7fffffff, rng.randrange(1 << 32)]))
        else:
            i = 12 + rng.randrange(100)
            b[i:i + 8] = struct.pack("<Q", rng.choice([0, 1, len(b), 1 << 40, (1 << 64) - 1]))
        return bytes(b)
    codes = set()
    for _ in range(600):
        m = mutate(r, len(r))
        for fn in (lambda x: amd.r1cs_setup(x, 1, 2), lambda x: amd.plonk_setup(x, 1, device=0)):
            try:
                fn(m)
            except amd.G16Error as e:
                codes.add(e.value.code if hasattr(e, "value") else e.code)
    for _ in range(400):
        try:
            amd.plonk_setup_ptau(r, mutate(pt, 400), device=0)
        except amd.G16Error as e:
            codes.add(e.code)
    assert codes <= {-1, -2, -4, -5}

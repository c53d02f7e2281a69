"""SURVEY 8f row 2 on the device: the trapdoor setup's fixed-base multiplications on the GPU (csrc/setup_gpu.hip,
selected with g16_setup_device) must give the byte-identical .zkey / verification key that the host path, the Python
oracle setup (oracle/groth16.py) and the committed golden fixture give -- for the synthetic circuits, for an iden3
.r1cs read back by g16_r1cs_setup, and for the natively built NZCP example circuit (every section: A, B1, B2, C, H, IC)."""
import hashlib
import json
import time

import pytest

import formats as f
import groth16 as g
import synth
from conftest import golden_path

pytestmark = pytest.mark.gpu


@pytest.fixture()
def on_device(amd):
    amd.setup_device(0)
    yield amd
    amd.setup_device(-1)


@pytest.mark.parametrize("n,p,m,seed", [(150, 6, 120, 2), (333, 20, 300, 9), (1000, 513, 400, 5)])
def test_device_setup_equals_python_oracle(on_device, n, p, m, seed):
    amd = on_device
    zkey, wtns, vkey = amd.synth_setup(n, p, m, seed, 4)
    rows, w = synth.make(n, p, m, seed)
    zk, _ = g.setup(n, p, rows, g.trapdoor(seed + 1))
    assert zkey == f.write_zkey(zk)
    assert wtns == f.write_wtns(w)
    exp_vk = f.g1_to_lem(zk["alpha1"]) + f.g2_to_lem(zk["beta2"]) + f.g2_to_lem(zk["gamma2"]) + \
        f.g2_to_lem(zk["delta2"]) + b"".join(f.g1_to_lem(P) for P in zk["IC"])
    assert vkey == exp_vk


def test_device_setup_reproduces_golden(on_device):
    amd = on_device
    meta = json.load(open(golden_path("small.json")))
    zkey, wtns, _ = amd.synth_setup(meta["n"], meta["p"], meta["m"], meta["seed"], 2)
    assert zkey == open(golden_path("small.zkey"), "rb").read()
    assert wtns == open(golden_path("small.wtns"), "rb").read()


def test_device_r1cs_setup_equals_host(amd):
    n, p, m, seed = 5000, 7, 4100, 13
    _, rows, _ = synth.gen_circuit(n, p, m, seed)
    r1cs = f.write_r1cs(n, p, 0, rows)
    zkey_h, vkey_h = amd.r1cs_setup(r1cs, seed, 8)
    amd.setup_device(0)
    try:
        zkey_d, vkey_d = amd.r1cs_setup(r1cs, seed, 8)
    finally:
        amd.setup_device(-1)
    assert zkey_d == zkey_h and vkey_d == vkey_h


def test_device_setup_at_scale_equals_host_and_proves(amd):
    """nzcp_live-shaped synthetic circuit, 300 k wires (domain 2^19): host and device keys byte-identical (chunked
    device path: more than one chunk of 2^20 scalars only at full size, exercised by bench.py), and the key proves."""
    n, p, m, seed = 300_000, 513, 300_000, 77
    t0 = time.time()
    zkey_h, wtns, vkey_h = amd.synth_setup(n, p, m, seed, 0)
    t1 = time.time()
    amd.setup_device(0)
    try:
        zkey_d, _, vkey_d = amd.synth_setup(n, p, m, seed, 0)
    finally:
        amd.setup_device(-1)
    t2 = time.time()
    print(f"setup n={n}: host {t1 - t0:.2f}s, device {t2 - t1:.2f}s")
    assert hashlib.sha256(zkey_d).digest() == hashlib.sha256(zkey_h).digest()
    assert vkey_d == vkey_h
    prover = amd.Prover(zkey_d, device=0)
    proof, pub = prover.prove(wtns)
    prover.close()
    vk = {"alpha1": f.g1_from_lem(vkey_d[0:64]), "beta2": f.g2_from_lem(vkey_d[64:192]),
          "gamma2": f.g2_from_lem(vkey_d[192:320]), "delta2": f.g2_from_lem(vkey_d[320:448]),
          "IC": [f.g1_from_lem(vkey_d[448 + 64 * i:512 + 64 * i]) for i in range(p + 1)]}
    pts = (f.g1_from_obj(proof["pi_a"]), f.g2_from_obj(proof["pi_b"]), f.g1_from_obj(proof["pi_c"]))
    assert g.verify(vk, [int(x) for x in pub], pts)

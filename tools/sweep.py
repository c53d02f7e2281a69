#!/usr/bin/env python3
"""Tuning sweep on the GPU box: one synthetic key, many prover configurations (window bits per MSM,
task length) via the G16_WINDOW_BITS / G16_TASK_LEN overrides.  Prints per-phase ms per config.
usage: python tools/sweep.py [n_vars] < configs (one 'a,b1,b2,c,h[;task_len]' per line)"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import __graft_entry__ as entry  # noqa: E402

amd = entry.load_package()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_700_000
zkey, wtns, _ = amd.synth_setup(n, 513, n, 0x6E7A6370, 0)
r = (12345).to_bytes(32, "little")
s = (67890).to_bytes(32, "little")
ref = None
for line in sys.stdin:
    line = line.strip()
    if not line or line.startswith("#"):
        continue
    wb, _, tl = line.partition(";")
    os.environ["G16_WINDOW_BITS"] = wb
    if tl:
        os.environ["G16_TASK_LEN"] = tl
    else:
        os.environ.pop("G16_TASK_LEN", None)
    pv = amd.Prover(zkey)
    pv.stage(0, wtns)
    pr, pub = amd.Proof(), ctypes.create_string_buffer(513 * 32)
    pv.prove_staged_raw(0, r, s, pr, pub)
    t0 = time.perf_counter()
    K = 3
    acc = [0.0] * 5
    accum = [0.0] * 5
    for _ in range(K):
        assert pv.prove_staged_raw(0, r, s, pr, pub) == 0
        tm = pv.timings()
        for i in range(5):
            acc[i] += tm["msm_ms"][i] / K
            accum[i] += tm["msm_accum_kernel_ms"][i] / K
    dt = (time.perf_counter() - t0) / K * 1e3
    b = bytes(pr.a) + bytes(pr.b) + bytes(pr.c)
    if ref is None:
        ref = b
    print(f"{line:28s} c={list(pv.info.window_bits)} ms/proof={dt:7.2f} msm={[round(x, 2) for x in acc]} "
          f"accum={[round(x, 2) for x in accum]} same_proof={b == ref}", flush=True)
    pv.close()

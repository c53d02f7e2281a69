#!/usr/bin/env python3
"""bench.py -- Groth16 proofs/s on the nzcp_live-shaped R1CS (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU, backend nccl = RCCL)

A "step" is ONE Groth16 proof of the workload circuit through the C ABI (g16_prove_staged: QAP
evaluation -> 6 NTTs -> join -> 5 Pippenger MSMs -> blinding -> affine proof bytes), the witness
already resident in HBM.  N = 1: whole key on one GPU.  N > 1 (BASELINE config 4): every base
section is sharded by contiguous point range across the ranks (g16_opts.shard_*), each rank runs
its partial MSMs, the 768-byte partial-sum blobs are exchanged with an RCCL all-gather and every
rank finishes the proof -- total work is fixed, so scaling is "strong".  `--mode replicas` runs N
independent provers instead (weak scaling, batch mode of config 3).

Workload: the real nzcp_live R1CS cannot be built offline (no circom), so this is the
shape-matched synthetic circuit of SURVEY 8d config 2: nVars = nConstraints = 1.7 M (upper
structural estimate, domain 2^21), 513 public signals, NZCP witness value mix; trapdoor setup by
the product's own g16_synth_setup.  `data` says so.

Extra objects on the JSON line:
  roofline     -- the dominant kernel (msm_accumulate_kernel<G1>): algorithmic bytes
                  (points x 96 B) / HIP-event kernel time, vs 8 TB/s HBM.  The kernel is
                  integer-VALU bound by construction; the fraction is reported as the metric asks.
  cpu_baseline -- the oracle's C restatement (oracle/c, OpenMP) timed on this box's host cores on
                  a 1/8-size sample of the same workload, scaled to proofs/s; also checks the GPU
                  proof of that sample bit-for-bit and pairing-verifies the full-size proof.
"""
import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import __graft_entry__ as entry  # noqa: E402

SEED = 0x6E7A6370
HBM_PEAK_GBS = 8000.0


def fixed_rs(seed):
    """Deterministic blinding (seed+2 stream of the synthetic spec) so runs are reproducible."""
    MASK = (1 << 64) - 1
    R = 21888242871839275222246405745257275088548364400416034343698204186575808495617

    def splitmix(z):
        out = []
        for _ in range(4):
            z = (z + 0x9E3779B97F4A7C15) & MASK
            x = z
            x = ((x ^ (x >> 30)) * 0xBF58476D1CE4E5B9) & MASK
            x = ((x ^ (x >> 27)) * 0x94D049BB133111EB) & MASK
            out.append(x ^ (x >> 31))
        return out
    s = splitmix(seed + 2)

    def rotl(x, k):
        return ((x << k) | (x >> (64 - k))) & MASK

    def nxt():
        res = (rotl((s[1] * 5) & MASK, 7) * 9) & MASK
        t = (s[1] << 17) & MASK
        s[2] ^= s[0]; s[3] ^= s[1]; s[1] ^= s[2]; s[0] ^= s[3]
        s[2] ^= t
        s[3] = rotl(s[3], 45)
        return res

    def fr():
        u = [nxt() for _ in range(4)]
        return (u[0] | (u[1] << 64) | (u[2] << 128) | (u[3] << 192)) % R
    r, s_ = fr(), fr()
    return r.to_bytes(32, "little"), s_.to_bytes(32, "little")


def cpu_leg(amd, args, full_proof, full_pub, full_vkey, full_zkey, full_wtns, full_gpu_bytes, log):
    """cpu_baseline (rank 0, N = 1): time the oracle's C prover on a bounded sample, check the GPU
    against it bit-for-bit on that sample, pairing-verify the full-size GPU proof."""
    entry.oracle_path()
    lib_path = os.path.join(ROOT, "oracle", "_build", "libg16oracle.so")
    if not os.path.exists(lib_path):
        return {"value": None, "unit": "proofs/s", "cores": 0, "kind": "port",
                "sample": "oracle/_build/libg16oracle.so not built"}
    olib = ctypes.CDLL(lib_path)
    olib.g16o_prove.argtypes = [ctypes.c_char_p, ctypes.c_size_t, ctypes.c_char_p, ctypes.c_size_t,
                                ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p, ctypes.c_char_p,
                                ctypes.c_int]
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except Exception:
        pass
    div = args.cpu_sample_div
    out = ctypes.create_string_buffer(256)
    pub = ctypes.create_string_buffer(max(1, args.n_public * 32))
    if div <= 1:
        # the sample IS the workload: same zkey, witness and blinding as the timed GPU steps
        div, n_s = 1, args.n_vars
        r, s = fixed_rs(SEED)
        t0 = time.time()
        rc = olib.g16o_prove(full_zkey, len(full_zkey), full_wtns, len(full_wtns), r, s, out, pub, cores)
        t_cpu = time.time() - t0
        assert rc == 0, "oracle prover failed"
        sample_ok = (full_gpu_bytes == out.raw)
    else:
        n_s, m_s = max(args.n_vars // div, 600), max(args.n_constraints // div, 64)
        zk, wt, _ = amd.synth_setup(n_s, args.n_public, m_s, SEED + 7, 0)
        r, s = fixed_rs(SEED + 7)
        t0 = time.time()
        rc = olib.g16o_prove(zk, len(zk), wt, len(wt), r, s, out, pub, cores)
        t_cpu = time.time() - t0
        assert rc == 0, "oracle prover failed"
        pv = amd.Prover(zk, device=0)
        pr, gpub = amd.Proof(), ctypes.create_string_buffer(max(1, args.n_public * 32))
        pv.stage(0, wt)
        assert pv.prove_staged_raw(0, r, s, pr, gpub) == 0
        gpu_bytes = bytes(pr.a) + bytes(pr.b) + bytes(pr.c)
        sample_ok = (gpu_bytes == out.raw and gpub.raw == pub.raw)
        pv.close()
    log(f"cpu sample n={n_s}: {t_cpu:.2f}s on {cores} threads; GPU==CPU proof bytes: {sample_ok}")
    assert sample_ok, "GPU proof differs from the CPU oracle on the sample circuit"
    # pairing check of the full-size GPU proof against the verification key of the setup
    import formats as f
    import groth16 as g
    p = args.n_public
    vk = {"alpha1": f.g1_from_lem(full_vkey[0:64]), "beta2": f.g2_from_lem(full_vkey[64:192]),
          "gamma2": f.g2_from_lem(full_vkey[192:320]), "delta2": f.g2_from_lem(full_vkey[320:448]),
          "IC": [f.g1_from_lem(full_vkey[448 + 64 * i:512 + 64 * i]) for i in range(p + 1)]}
    pts = (f.g1_from_obj(full_proof["pi_a"]), f.g2_from_obj(full_proof["pi_b"]), f.g1_from_obj(full_proof["pi_c"]))
    verified = g.verify(vk, [int(x) for x in full_pub], pts)
    log(f"full-size proof pairing check: {verified}")
    assert verified, "full-size GPU proof failed the pairing check"
    return {"value": round(1.0 / (t_cpu * div), 5), "unit": "proofs/s", "cores": cores, "kind": "port",
            "sample": (f"oracle/c OpenMP prover (in-repo CPU port, not snarkjs), one proof of "
                       + ("the full workload" if div == 1 else f"the 1/{div}-size circuit, scaled x{div} linearly")
                       + f" (nVars={n_s}) in {t_cpu:.2f}s on {cores} threads; GPU proof bit-identical to it; "
                         f"full-size GPU proof pairing-verified"),
            "sample_seconds": round(t_cpu, 3)}


def batch_leg(amd, args, zkey, wtns, prover0, r, s, log):
    """Throughput mode (BASELINE config 3, "batch of independent witnesses, 1-GPU throughput mode"):
    g16_prove_batch on ONE resident handle -- the library software-pipelines the batch over three
    per-proof scratch contexts (proofs i+1, i+2 on the GPU while the host collects and finishes proof i).
    The witnesses come from host memory, so this figure INCLUDES the PCIe upload of every witness.
    Reported next to the single-proof `value`, never instead of it."""
    nslots = min(8, max(1, args.batch_proofs))
    if getattr(args, "nz", None):   # the real circuit: witnesses of other passes (same constraint system)
        nz_params, nz_pass = args.nz
        wts = [wtns] + [amd.nzcp_circuit_setup(nz_params, nz_pass(i), SEED, 0, want_zkey=False)["wtns"] for i in range(1, nslots)]
    else:
        wts = [wtns] + [amd.synth_witness(args.n_vars, args.n_public, args.n_constraints, SEED, SEED + 100 + i)
                        for i in range(1, nslots)]
    total = args.batch_proofs
    lib = amd.load()
    arr = (ctypes.c_char_p * total)(*[wts[i % nslots] for i in range(total)])
    lens = (ctypes.c_size_t * total)(*[len(wts[i % nslots]) for i in range(total)])
    rs = (r + s) * total
    out = (amd.Proof * total)()
    pubs = ctypes.create_string_buffer(max(1, total * args.n_public * 32))
    # reference proofs of the distinct witnesses, one at a time
    ref = []
    pr, pub = amd.Proof(), ctypes.create_string_buffer(max(1, args.n_public * 32))
    for k in range(nslots):
        prover0.stage(1, wts[k])
        assert prover0.prove_staged_raw(1, r, s, pr, pub) == 0
        ref.append(bytes(pr.a) + bytes(pr.b) + bytes(pr.c))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = lib.g16_prove_batch(prover0._h, arr, lens, total, rs, out, pubs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert rc == 0, lib.g16_last_error()
    for i in range(total):
        assert bytes(out[i].a) + bytes(out[i].b) + bytes(out[i].c) == ref[i % nslots], "batch proof differs from the single proof"
    log(f"batch throughput: {total} proofs by g16_prove_batch in {dt * 1e3:.1f} ms")
    res = {"proofs_per_sec": round(total / dt, 3), "proofs": total, "mode": "g16_prove_batch, 3 pipelined contexts, "
           "witnesses uploaded from host memory inside the timed region", "distinct_witnesses": nslots,
           "ms_per_proof": round(1e3 * dt / total, 3)}
    return res, bytes(out), pubs.raw[:total * args.n_public * 32]


def verify_leg(amd, args, vkey, proofs, pubs, dev, log):
    """SURVEY 8f row 4 / north_star "every proof verifying against the reference verification key": ALL proofs of the
    throughput leg are checked by the device batch verifier (g16_verify_batch: one verdict per proof) against the
    verification key of the setup; one deliberately corrupted copy must come back rejected at its index."""
    total = len(proofs) // 256
    ver = amd.Verifier(vkey, args.n_public, montgomery=True, device=dev)
    ver.verify_raw(proofs[:256 * min(total, 64)], pubs, min(total, 64))     # warm-up
    best, phases = None, None
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ok = ver.verify_raw(proofs, pubs, total)
        dt = time.perf_counter() - t0
        if best is None or dt < best:
            best, phases = dt, ver.timings()
    assert all(ok), f"device verifier rejected proofs {[i for i, o in enumerate(ok) if not o][:8]}"
    k = total // 2
    bad = bytearray(pubs)
    bad[(k * args.n_public) * 32] ^= 1
    ok2 = ver.verify_raw(proofs, bytes(bad), total)
    assert ok2 == [i != k for i in range(total)], "device verifier missed the corrupted statement"
    ver.close()
    log(f"device verifier: {total} proofs in {best * 1e3:.1f} ms, all accepted; corrupted statement {k} rejected")
    return {"verifications_per_sec": round(total / best, 1), "proofs": total, "ms": round(best * 1e3, 3),
            "device_phases_ms": {"vk_x": round(phases[0], 3), "miller_loops": round(phases[1], 3), "final_exp": round(phases[2], 3)},
            "all_accepted": True, "corrupted_statement_rejected": True,
            "mode": "g16_verify_batch: host buffers in, one verdict per proof out (uploads inside the timed region)"}


def plonk_leg(amd, dev, log, steps=3):
    """the PLONK prover (csrc/plonk.hip) on nzcp_exampleTest as PLONK -- the circuit whose PLONK setup
    the reference scripts (Makefile:30-33).  Key from the test-only setup with a known tau; fresh random blinding per
    proof; the last proof is checked by the oracle's KZG verifier.  An extra figure, never `value`."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    entry.oracle_path()
    import nzcp_pass
    import plonk as pk
    t0 = time.time()
    params = amd.NZCP_EXAMPLE_PARAMS
    out = amd.nzcp_circuit_setup(params, nzcp_pass.to_be_signed("Jack", "Sparrow", "1960-04-16", live=False, exp=1951416330),
                                 SEED, 0, want_zkey=False, want_r1cs=True)
    zkey = amd.plonk_setup(out["r1cs"], SEED, device=dev, with_lagrange=False)
    vk = pk.vkey_from_zkey(zkey)
    prover = amd.PlonkProver(zkey, device=dev)
    del zkey
    prover.prove_raw(out["wtns"])
    ts = []
    for _ in range(steps):
        t = time.perf_counter()
        prover.prove_raw(out["wtns"])
        ts.append(time.perf_counter() - t)
    rounds = prover.timings()
    proof, pub = prover.prove(out["wtns"])
    ok = pk.verify(vk, [int(x) for x in pub], pk.proof_from_obj(proof))
    # ... and by the device PLONK verifier (csrc/verify_plonk.hip): 256 copies, one with a changed public signal

    def g1j(P):
        return ["0", "1", "0"] if P is None else [str(P[0]), str(P[1]), "1"]
    vkj = {"protocol": "plonk", "nPublic": vk["nPublic"], "power": vk["power"], "k1": vk["k1"], "k2": vk["k2"],
           "X_2": [[str(vk["X_2"][0][0]), str(vk["X_2"][0][1])], [str(vk["X_2"][1][0]), str(vk["X_2"][1][1])], ["1", "0"]]}
    for k_ in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
        vkj[k_] = g1j(vk[k_])
    pver = amd.PlonkVerifier(vkj, device=dev)
    bad_pub = list(pub)
    bad_pub[0] = str(1 - int(bad_pub[0])) if bad_pub[0] in ("0", "1") else str(int(bad_pub[0]) + 1)
    items = [(pub, proof)] * 255 + [(bad_pub, proof)]
    pver.verify_batch(items[:8])
    tv = time.perf_counter()
    verdicts = pver.verify_batch(items)
    tv = time.perf_counter() - tv
    pver.close()
    dev_ok = verdicts == [True] * 255 + [False]
    res = {"proofs_per_sec": round(1 / min(ts), 3), "ms_per_proof": round(min(ts) * 1e3, 2),
           "device_verifier": {"proofs": 256, "ms": round(tv * 1e3, 1), "verdicts_as_expected": bool(dev_ok)},
           "workload": f"nzcp_exampleTest as PLONK: domain 2^{prover.domain_size.bit_length() - 1}, {prover.n_constraints} gates, "
                       f"{prover.n_additions} addition gates, {prover.n_public} public signals",
           "rounds_ms": rounds, "verified_by_oracle_kzg": bool(ok), "prepare_s": round(time.time() - t0, 1)}
    prover.close()
    log(f"plonk: {res['ms_per_proof']} ms per proof, verified {ok}")
    assert ok, "PLONK proof rejected by the oracle verifier"
    assert dev_ok, "device PLONK verifier: unexpected verdicts"
    return res


def shard_rehearsal_leg(amd, args, zkey, wtns, prover, r, s, want_bytes, dev, log):
    """BASELINE config 4 rehearsed on ONE device (the driver's 8-GPU node runs the real thing): for G = 2, 4, 8 every
    shard handle of the G-way point-range split is created on this GPU and timed ALONE through the two-phase C ABI
    (g16_shard_begin: QAP + odd-coset evaluation of the vectors the rank owns, witness MSMs of its point range;
    g16_shard_end: its slices of A, B, C -> join -> H-MSM of its range -> partial sums).  A G-GPU proof's critical path
    is max over the owner ranks (begin) + the slice exchange + max over ranks (end); the partial sums of all G ranks
    are finished on the host and must equal the unsharded proof bytes."""
    N, EB = prover.info.domain_size, amd.LAZY_FR_BYTES
    full = [torch.zeros(N * EB, dtype=torch.uint8, device="cuda") for _ in range(3)]
    prover.shard_begin(0, 7, [t.data_ptr() for t in full])       # the three coset evaluations, once
    prover.shard_end(0, [t.data_ptr() for t in full])             # (lo = 0: completes the unsharded handle's proof)
    torch.cuda.synchronize()
    res = {}
    for G in (2, 4, 8):
        parts, begin_ms, end_ms = [], [], []
        for rank in range(G):
            sh = amd.Prover(zkey, device=dev, shard_rank=rank, shard_count=G, window_bits=args.window_bits, task_len=args.task_len)
            sh.stage(0, wtns)
            mask = sum(1 << v for v in range(3) if amd.shard_vector_owner(v, G) == rank)
            own = [torch.zeros(N * EB, dtype=torch.uint8, device="cuda") if (mask >> v) & 1 else None for v in range(3)]
            lo, hi = amd.shard_range(N, rank, G)
            best_b = best_e = None
            for _ in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                sh.shard_begin(0, mask, [t.data_ptr() if t is not None else 0 for t in own])
                t1 = time.perf_counter()
                blob = sh.shard_end(0, [t.data_ptr() + lo * EB for t in full])
                t2 = time.perf_counter()
                best_b = min(best_b, t1 - t0) if best_b else t1 - t0
                best_e = min(best_e, t2 - t1) if best_e else t2 - t1
            for v in range(3):
                if own[v] is not None:
                    assert torch.equal(own[v], full[v]), "a shard's coset evaluation differs from the unsharded one"
            parts.append(blob)
            begin_ms.append(round(best_b * 1e3, 3))
            end_ms.append(round(best_e * 1e3, 3))
            sh.close()
            del own
        got = amd.finish_host(zkey, parts, r, s)
        assert got == want_bytes, f"{G}-shard proof differs from the unsharded proof"
        res[f"G{G}"] = {"begin_ms_by_rank": begin_ms, "end_ms_by_rank": end_ms,
                        "critical_path_ms_excl_exchange": round(max(begin_ms) + max(end_ms), 3),
                        "proof_equals_unsharded": True}
        log(f"shard rehearsal G={G}: begin {begin_ms} end {end_ms}")
    res["note"] = ("every shard handle timed alone on this one GPU (host wall time around g16_shard_begin / g16_shard_end, best of 3): "
                   "begin = QAP + coset evaluation of the owned vectors (ranks 0..2), the witness MSMs keep running; end = join + H-MSM "
                   "over the rank's range + wait for its witness MSMs + fold.  A G-GPU proof = max(begin) + slice exchange over xGMI "
                   "(3 x 40 B x N / G per rank) + max(end); not a multi-GPU measurement")
    return res


def upper_bracket_leg(amd, args, dev, threads, log):
    """SURVEY App. E cannot decide between 0.85 M and 1.7 M constraints for the circom-compiled nzcp_liveTest (10 or 39
    SHA-256 compressions inside Sha256Var): `value` is measured on the native restatement of the circuit (830 k rows, N = 2^20);
    this leg measures the SAME run's prover on the upper estimate -- the shape-matched synthetic circuit at 1.7 M
    constraints, N = 2^21 (round 1's headline workload) -- single proof and throughput mode."""
    n = 1_700_000
    t0 = time.time()
    zkey, wtns, vkey = amd.synth_setup(n, 513, n, SEED, threads)
    pv = amd.Prover(zkey, device=dev, window_bits=args.window_bits, task_len=args.task_len)
    del zkey
    log(f"upper bracket: synthetic nVars=nConstraints={n}, setup+create {time.time() - t0:.1f}s")
    pv.stage(0, wtns)
    r, s = fixed_rs(SEED)
    pr, pub = amd.Proof(), ctypes.create_string_buffer(513 * 32)
    for _ in range(3):
        assert pv.prove_staged_raw(0, r, s, pr, pub) == 0
    K = 10
    acc = {"qap_ms": 0.0, "ntt_ms": 0.0, "total_ms": 0.0}
    msm = [0.0] * 5
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(K):
        assert pv.prove_staged_raw(0, r, s, pr, pub) == 0
        tm = pv.timings()
        for k_ in acc:
            acc[k_] += tm[k_]
        for i in range(5):
            msm[i] += tm["msm_ms"][i]
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / K
    single = bytes(pr.a) + bytes(pr.b) + bytes(pr.c)
    # throughput mode: 64 proofs over 4 distinct witnesses, uploads inside
    nslots, total = 4, 64
    wts = [wtns] + [amd.synth_witness(n, 513, n, SEED, SEED + 100 + i) for i in range(1, nslots)]
    lib = amd.load()
    arr = (ctypes.c_char_p * total)(*[wts[i % nslots] for i in range(total)])
    lens = (ctypes.c_size_t * total)(*[len(wts[i % nslots]) for i in range(total)])
    out = (amd.Proof * total)()
    pubs = ctypes.create_string_buffer(total * 513 * 32)
    assert lib.g16_prove_batch(pv._h, arr, lens, 8, (r + s) * 8, out, pubs) == 0      # warm-up
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = lib.g16_prove_batch(pv._h, arr, lens, total, (r + s) * total, out, pubs)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert rc == 0, lib.g16_last_error()
    assert bytes(out[0].a) + bytes(out[0].b) + bytes(out[0].c) == single, "batch proof differs from the single proof"
    pv.close()
    ver = amd.Verifier(vkey, 513, montgomery=True, device=dev)
    ok = ver.verify_raw(bytes(out), pubs.raw, total)
    ver.close()
    assert all(ok), "device verifier rejected a proof of the upper-bracket circuit"
    return {"workload": f"nzcp_live-shaped synthetic R1CS at SURVEY App. E's UPPER estimate: nVars=nConstraints={n}, nPublic=513, domain=2^21",
            "ms_per_proof": round(ms, 3), "proofs_per_sec": round(1e3 / ms, 2),
            "phases_ms": {"qap": round(acc["qap_ms"] / K, 3), "ntt_x6_join": round(acc["ntt_ms"] / K, 3),
                          "msm_A_B1_B2_C_H": [round(x / K, 3) for x in msm], "device_total": round(acc["total_ms"] / K, 3)},
            "batch": {"proofs": total, "distinct_witnesses": nslots, "proofs_per_sec": round(total / dt, 2),
                      "ms_per_proof": round(1e3 * dt / total, 3), "all_verified_on_device": True}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-vars", type=int, default=1_700_000)
    ap.add_argument("--n-constraints", type=int, default=1_700_000)
    ap.add_argument("--n-public", type=int, default=513)
    ap.add_argument("--circuit", choices=["nzcp_live", "nzcp_example", "synthetic"], default="nzcp_live",
                    help="workload circuit: the REAL NZCPPubIdentity constraint system built natively with the CBOR search "
                         "in the circuit (nzcp_live = nzcp_liveTest.circom's parameters on a live-format pass, BASELINE "
                         "configs 2-4; nzcp_example = nzcp_exampleTest.circom on the MoH example pass, config 1), or the "
                         "shape-matched synthetic R1CS at --n-vars / --n-constraints (round 1's upper structural estimate)")
    ap.add_argument("--sha256-blocks", type=int, default=0,
                    help="BASELINE config 5 on a REAL constraint system: this many chained SHA-256 compressions "
                         "(163 fill the 2^22 domain) instead of the shape-matched synthetic circuit")
    ap.add_argument("--mode", choices=["shard", "replicas"], default="shard")
    ap.add_argument("--window-bits", type=int, default=0)
    ap.add_argument("--task-len", type=int, default=0)
    ap.add_argument("--cpu-sample-div", type=int, default=1)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--batch-streams", type=int, default=2,
                    help="extra measurement at N=1: throughput mode of BASELINE config 3 (g16_prove_batch over "
                         "independent witnesses); 0 = skip")
    ap.add_argument("--batch-proofs", type=int, default=1024,
                    help="proofs in the throughput leg: BASELINE config 3 is stated on 1 024 independent witnesses "
                         "(8 distinct ones cycled; every proof is compared with its one-by-one result)")
    ap.add_argument("--no-plonk", action="store_true", help="skip the PLONK prover leg (extra figure at N=1)")
    ap.add_argument("--no-brackets", action="store_true",
                    help="skip the upper_bracket leg (1.7 M-constraint synthetic circuit) and the shard_rehearsal leg (N=1)")
    args = ap.parse_args()

    args.nz = None
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    assert torch.cuda.is_available(), "bench.py needs a GPU (no CPU fallback)"
    # Rehearsal knobs (the driver never sets them): G16_BENCH_DEVICE pins every rank to one GPU and
    # G16_BENCH_BACKEND=gloo exchanges the partial sums over gloo, so the N > 1 code path can be
    # exercised end to end on a 1-GPU box.  Default: one rank per GPU, RCCL ("nccl") over xGMI.
    dev = int(os.environ.get("G16_BENCH_DEVICE", local))
    backend = os.environ.get("G16_BENCH_BACKEND", "nccl")
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(backend)
    xdev = "cuda" if backend == "nccl" else "cpu"

    def log(msg):
        if rank == 0:
            print(f"[bench] {msg}", file=sys.stderr, flush=True)

    amd = entry.load_package()
    amd.load()
    ncpu = os.cpu_count() or 1
    threads = max(1, ncpu // world)
    # the trapdoor setup's fixed-base multiplications run on this rank's GPU (csrc/setup_gpu.hip; same key bytes as the
    # host-thread path, tests/test_gpu_setup.py) -- untimed preparation either way
    amd.setup_device(dev)
    t0 = time.time()
    if args.sha256_blocks > 0:
        import hashlib
        out = amd.sha256_chain_setup(args.sha256_blocks, hashlib.sha256(b"nzcp-circom config 5").digest(), SEED, threads)
        zkey, wtns, vkey = out["zkey"], out["wtns"], out["vkey"]
        args.n_public = 256
        args.n_vars = (len(wtns) - 76) // 32
        args.n_constraints = -1          # read back from the handle below
        args.batch_streams = 0           # the batch leg draws fresh witnesses from the synthetic generator
        args.cpu_sample_div = max(args.cpu_sample_div, 1)
        log(f"sha256-chain setup blocks={args.sha256_blocks} nVars={args.n_vars}: {time.time() - t0:.1f}s, "
            f"zkey {len(zkey) / 1e6:.0f} MB")
    elif args.circuit != "synthetic":
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import nzcp_pass
        live = args.circuit == "nzcp_live"
        nz_params = amd.NZCP_LIVE_PARAMS if live else amd.NZCP_EXAMPLE_PARAMS

        def nz_pass(i):   # distinct passes of the circuit's format (real live passes are private; the example pass is one)
            names = [("Jack", "Sparrow", "1960-04-16"), ("Anne-Marie", "Te Whare", "1987-11-30"), ("Li", "Wei", "2001-02-03"),
                     ("Aroha", "Ngata", "1975-07-21"), ("Sione", "Tuilagi", "1990-12-01"), ("Mere", "Hohepa", "1968-03-15"),
                     ("Tama", "Parata", "1983-09-09"), ("Olivia", "Smith", "1999-05-27")]
            if not live:   # nzcp_exampleTest's MaxToBeSignedBytes = 314 is the example pass's own length: no longer names
                names = [n for n in names if len(n[0]) + len(n[1]) <= 11]
            g_, f_, d_ = names[i % len(names)]
            return nzcp_pass.to_be_signed(g_, f_, d_, live=live, exp=1951416330 - i)
        out = amd.nzcp_circuit_setup(nz_params, nz_pass(0), SEED, threads)
        zkey, wtns, vkey = out["zkey"], out["wtns"], out["vkey"]
        args.n_public = 513
        args.n_vars = (len(wtns) - 76) // 32
        args.n_constraints = out["n_constraints"]
        args.nz = (nz_params, nz_pass)
        log(f"{args.circuit}: NZCPPubIdentity{tuple(nz_params)} built natively with the CBOR search in the circuit: "
            f"{args.n_constraints} constraints, {args.n_vars} wires; setup {time.time() - t0:.1f}s, zkey {len(zkey) / 1e6:.0f} MB")
    else:
        zkey, wtns, vkey = amd.synth_setup(args.n_vars, args.n_public, args.n_constraints, SEED, threads)
        log(f"synthetic setup nVars={args.n_vars} nConstraints={args.n_constraints}: {time.time() - t0:.1f}s, "
            f"zkey {len(zkey) / 1e6:.0f} MB")
    sharded = world > 1 and args.mode == "shard"
    t0 = time.time()
    prover = amd.Prover(zkey, device=dev, shard_rank=rank if sharded else 0,
                        shard_count=world if sharded else 1, window_bits=args.window_bits,
                        task_len=args.task_len)
    zkey_r = zkey if sharded else None      # kept for the replicas measurement below
    if world > 1 or (args.no_cpu and args.no_brackets):
        zkey = None
    info = prover.info
    log(f"g16_create: {time.time() - t0:.1f}s; domain 2^{info.domain_size.bit_length() - 1}, nCoefs {info.n_coefs}, "
        f"resident bases A/B1/B2/C/H = {info.n_a}/{info.n_b1}/{info.n_b2}/{info.n_c}/{info.n_h}, "
        f"window bits {list(info.window_bits)}")
    prover.stage(0, wtns)
    upload_ms = prover.timings()["upload_ms"]
    r, s = fixed_rs(SEED)
    pr = amd.Proof()
    pub = ctypes.create_string_buffer(max(1, args.n_public * 32))
    gather_in = torch.zeros(amd.PARTIAL_BYTES, dtype=torch.uint8, device=xdev)
    gather_out = torch.zeros(amd.PARTIAL_BYTES * world, dtype=torch.uint8, device=xdev)
    lib = amd.load()
    pbuf = ctypes.create_string_buffer(amd.PARTIAL_BYTES)
    # Sharded H pipeline (g16_shard_begin / g16_shard_end): rank v mod N evaluates vector v of (A, B, C) on the
    # coset, a scatter per vector hands every rank its slice [lo, hi) of the domain (RCCL over xGMI; equal-size
    # chunks, padded), each rank joins and multi-exponentiates only its own range of P.
    EB = amd.LAZY_FR_BYTES
    N_dom = info.domain_size
    pad = max(amd.shard_range(N_dom, k, world)[1] - amd.shard_range(N_dom, k, world)[0] for k in range(world)) if sharded else 0
    my_mask = sum(1 << v for v in range(3) if amd.shard_vector_owner(v, world) == rank) if sharded else 0
    vec_t = [torch.zeros((N_dom + pad) * EB, dtype=torch.uint8, device=xdev) if (my_mask >> v) & 1 else None
             for v in range(3)] if sharded else []
    slice_t = [torch.zeros(max(1, pad) * EB, dtype=torch.uint8, device=xdev) for _ in range(3)] if sharded else []

    def step():
        if not sharded:
            rc = prover.prove_staged_raw(0, r, s, pr, pub)
            assert rc == 0, lib.g16_last_error()
            return
        prover.shard_begin(0, my_mask, [vec_t[v].data_ptr() if vec_t[v] is not None else 0 for v in range(3)])
        for v in range(3):
            owner = amd.shard_vector_owner(v, world)
            chunks = None
            if rank == owner:
                chunks = []
                for k in range(world):
                    lo_k = amd.shard_range(N_dom, k, world)[0]
                    chunks.append(vec_t[v][lo_k * EB:(lo_k + max(1, pad)) * EB])
            dist.scatter(slice_t[v], chunks, src=owner)
        if xdev == "cuda":
            torch.cuda.synchronize()
        blob_mine = prover.shard_end(0, [t.data_ptr() for t in slice_t])
        gather_in.copy_(torch.frombuffer(bytearray(blob_mine), dtype=torch.uint8))
        dist.all_gather_into_tensor(gather_out, gather_in)      # RCCL over xGMI: 768 B per rank
        blob = gather_out.cpu().numpy().tobytes()
        rc = lib.g16_prove_finish(prover._h, 0, blob, world, r, s, ctypes.byref(pr), pub)
        assert rc == 0, lib.g16_last_error()

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    acc = {"qap_ms": 0.0, "ntt_ms": 0.0, "total_ms": 0.0, "msm_ms": [0.0] * 5, "accum": [0.0] * 5}
    step_ms = []
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        step()
        step_ms.append(1e3 * (time.perf_counter() - ts))
        tm = prover.timings()
        for k in ("qap_ms", "ntt_ms", "total_ms"):
            acc[k] += tm[k]
        for i in range(5):
            acc["msm_ms"][i] += tm["msm_ms"][i]
            acc["accum"][i] += tm["msm_accum_kernel_ms"][i]
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=xdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    K = args.steps
    proofs = K * (world if (world > 1 and not sharded) else 1)
    value = proofs / elapsed

    # N > 1, sharded run: also record the throughput reading (BASELINE config 3 across GPUs): every
    # rank proves `steps` independent proofs on an UNSHARDED handle, no collective -- weak scaling.
    replicas = None
    if sharded and zkey_r is not None:
        rp = amd.Prover(zkey_r, device=dev, window_bits=args.window_bits, task_len=args.task_len)
        zkey_r = None
        rp.stage(0, wtns)
        sharded_bytes = bytes(pr.a) + bytes(pr.b) + bytes(pr.c)   # the proof the sharded steps assembled
        pr_u = amd.Proof()
        assert rp.prove_staged_raw(0, r, s, pr_u, pub) == 0
        assert bytes(pr_u.a) + bytes(pr_u.b) + bytes(pr_u.c) == sharded_bytes, \
            "sharded proof differs from the unsharded proof of the same (witness, r, s)"
        for _ in range(args.warmup):
            assert rp.prove_staged_raw(0, r, s, pr, pub) == 0
        fence()
        t0r = time.perf_counter()
        for _ in range(args.steps):
            assert rp.prove_staged_raw(0, r, s, pr, pub) == 0
        fence()
        dtr = time.perf_counter() - t0r
        tr = torch.tensor([dtr], dtype=torch.float64, device=xdev)
        dist.all_reduce(tr, op=dist.ReduceOp.MAX)
        replicas = {"proofs_per_sec": round(world * args.steps / float(tr.item()), 3), "scaling": "weak",
                    "mode": f"{world} independent unsharded provers, one per GPU, no collective",
                    "sharded_proof_equals_unsharded": True}
        rp.close()
    # the ABI entry a snarkjs user hits: g16_prove = witness upload over PCIe + the same pipeline
    incl_upload = None
    if world == 1:
        for _ in range(2):
            assert lib.g16_prove(prover._h, wtns, len(wtns), r, s, ctypes.byref(pr), pub) == 0, lib.g16_last_error()
        t_u = []
        for _ in range(min(K, 10)):
            ts = time.perf_counter()
            assert lib.g16_prove(prover._h, wtns, len(wtns), r, s, ctypes.byref(pr), pub) == 0, lib.g16_last_error()
            t_u.append(1e3 * (time.perf_counter() - ts))
        t_u.sort()
        incl_upload = {"ms_per_proof_p50": round(t_u[len(t_u) // 2], 3), "proofs_per_sec": round(1e3 / t_u[len(t_u) // 2], 3),
                       "entry": "g16_prove(wtns in pageable host memory): upload + canonicity check + proof, no staging"}
    batch = verify = None
    if world == 1 and args.batch_streams > 0:
        batch, batch_proofs, batch_pubs = batch_leg(amd, args, None, wtns, prover, r, s, log)
        verify = verify_leg(amd, args, vkey, batch_proofs, batch_pubs, dev, log)
    rehearsal = None
    if world == 1 and not args.no_brackets and args.circuit != "synthetic" and zkey is not None:
        want = amd.proof_to_obj(pr)
        try:
            rehearsal = shard_rehearsal_leg(amd, args, zkey, wtns, prover, r, s, want, dev, log)
        except AssertionError:
            raise
        except Exception as e:  # noqa: BLE001  (an extra leg must not cost the headline line)
            rehearsal = {"error": repr(e)[:300]}
    if rank == 0:
        proof_obj = amd.proof_to_obj(pr)
        pub_list = [str(int.from_bytes(pub.raw[i * 32:(i + 1) * 32], "little")) for i in range(args.n_public)]
        # roofline of the dominant kernel: the G1 bucket-accumulate launches -- one over the fused witness group
        # (the resident points of A, B1 and C) and one over H
        launches = [(pts_, acc["accum"][i] / K) for pts_, i in ((info.n_a + info.n_b1 + info.n_c, 0), (info.n_h, 4))
                    if pts_ > 0]
        bytes_per_launch = sum(96.0 * n for n, _ in launches) / max(1, len(launches))
        ms_per_launch = sum(ms for _, ms in launches) / max(1, len(launches))
        achieved = bytes_per_launch / (ms_per_launch * 1e-3) / 1e9 if ms_per_launch > 0 else 0.0
        # HBM traffic of that kernel from the PMC passes committed under profiles/ (bench.py cannot collect counters
        # itself).  The summary records the hash of the kernel sources and the workload it was taken on: anything
        # else -- a kernel edited since, another circuit -- reports null with the reason instead of a stale figure.
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import kernel_costs
        src_hash = kernel_costs.kernel_source_hash()
        workload_id = (f"sha256x{args.sha256_blocks}" if args.sha256_blocks > 0 else
                       f"{args.circuit}:{args.n_vars}:{args.n_constraints}")
        traffic, traffic_note = None, "no PMC summary (profiles/r03_pmc_traffic.json) for this build and workload"
        try:
            pm = json.load(open(os.path.join(ROOT, "profiles", "r03_pmc_traffic.json")))
            if world != 1:
                traffic_note = "PMC summary is a 1-GPU profile"
            elif pm.get("kernel_src_sha256") != src_hash:
                traffic_note = "stale: profiles/r03_pmc_traffic.json was taken on other kernel sources (re-run tools/profile_round.sh)"
            elif pm.get("workload_id") != workload_id:
                traffic_note = f"profiles/r03_pmc_traffic.json was taken on workload {pm.get('workload_id')}, not {workload_id}"
            else:
                traffic = pm["avg_traffic_bytes_per_launch"]
                traffic_note = ("rocprofv3 FETCH_SIZE + WRITE_SIZE per launch, profiles/r03_pmc_traffic.json "
                                "(separate --pmc passes, raw counters: see its 'method')")
        except Exception:
            pass
        # static instruction counts of the accumulate kernel, generated at build time from the compiler's assembly
        costs = None
        try:
            costs = json.load(open(os.path.join(ROOT, "nzcp-circom_amd", "lib", "kernel_costs.json")))
            if costs.get("kernel_src_sha256") != src_hash:
                costs = None
        except Exception:
            pass
        # the integer-issue peak comes from the committed summary of tools/microbench.hip (measured on this pool's MI355X),
        # valid only for the microbenchmark source it was taken with
        peak_tmad, peak_note = None, "no profiles/r03_microbench_int_rates.json for tools/microbench.hip as it is now"
        try:
            import hashlib
            mb = json.load(open(os.path.join(ROOT, "profiles", "r03_microbench_int_rates.json")))
            if mb.get("src_sha256") == hashlib.sha256(open(os.path.join(ROOT, "tools", "microbench.hip"), "rb").read()).hexdigest():
                peak_tmad = mb["peak_Tmad_per_s"]
                peak_note = (f"{mb['cus']} CU x 4 SIMD x 64 lanes x {mb['clock_ghz']} GHz / {mb['cycles_per_v_mad_u64_u32']} cycles per "
                             "wave-instruction at 4 waves/SIMD (tools/microbench.hip -> profiles/r03_microbench_int_rates.json)")
        except Exception:
            pass
        nn, N, k = info.n_vars, info.domain_size, info.n_coefs
        b_proof = 44 * k + 384 * nn + 512 * N - 64 * (info.n_public + 1)
        out = {
            "metric": "groth16_proofs_per_sec_nzcp_live", "value": round(value, 4), "unit": "proofs/s",
            "n_gpus": world, "steps": K, "warmup": args.warmup, "ms_per_step": round(1e3 * elapsed / K, 3),
            "ms_per_step_p50_min": [round(sorted(step_ms)[len(step_ms) // 2], 3), round(min(step_ms), 3)],
            "higher_is_better": True, "scaling": "strong" if (sharded or world == 1) else "weak",
            "vs_baseline": None, "dtype": "u32x8 (256-bit Montgomery integers)",
            "data": ("real SHA-256-chain constraint system built natively (g16_sha256_chain_setup) + trapdoor zkey"
                     if args.sha256_blocks > 0 else
                     ("real NZCPPubIdentity constraint system (CBOR search + 2 variable-length SHA-256 in the circuit) built "
                      "natively from the reference's templates (circom itself is not runnable offline: the row count, 830 k, is "
                      "the native builder's, not circom's -- see upper_bracket for the 1.7 M estimate), synthetic "
                      + ("live-format pass" if args.circuit == "nzcp_live" else "example-format pass") + ", trapdoor zkey")
                     if args.circuit != "synthetic" else
                     "synthetic (shape-matched nzcp_live R1CS + trapdoor zkey; real circuit not buildable offline)"),
            "config": {"workload": (f"sha256 chain, {args.sha256_blocks} compressions (BASELINE config 5), single proof: "
                                    f"nVars={nn}, nPublic={info.n_public}, domain=2^{N.bit_length() - 1}, nCoefs={k}"
                                    if args.sha256_blocks > 0 else
                                    (f"{args.circuit}Test.circom = NZCPPubIdentity{tuple(args.nz[0])}, single proof: "
                                     f"nVars={nn}, nConstraints={args.n_constraints}, nPublic={info.n_public}, "
                                     f"domain=2^{N.bit_length() - 1}, nCoefs={k}")
                                    if args.circuit != "synthetic" else
                                    f"nzcp_live-shaped synthetic single proof: nVars={nn}, nConstraints={args.n_constraints}, "
                                    f"nPublic={info.n_public}, domain=2^{N.bit_length() - 1}, nCoefs={k}"),
                       "parallelism": ("1gpu" if world == 1 else
                                       (f"msm-point-range-shard{world}+abc-vector-split+scatter+allgather" if sharded
                                        else f"replicas{world}")),
                       "window_bits": list(info.window_bits)},
            "phases_ms": {"qap": round(acc["qap_ms"] / K, 3), "ntt_x6_join": round(acc["ntt_ms"] / K, 3),
                          "msm_A_B1_B2_C_H": [round(x / K, 3) for x in acc["msm_ms"]],
                          "device_total": round(acc["total_ms"] / K, 3),
                          "witness_upload_pcie": round(upload_ms, 3)},
            "algorithmic_bytes_per_proof": b_proof,
            # the HBM-side phases against the same roofline (SURVEY 8d byte formulas; phase times of the product schedule,
            # i.e. with the witness MSMs running beside them)
            "phase_rooflines": {
                "qap_eval": {"algorithmic_bytes": 44 * k + 32 * nn + 96 * N,
                             "GBps": round((44 * k + 32 * nn + 96 * N) / (acc["qap_ms"] / K * 1e-3) / 1e9, 1) if acc["qap_ms"] else None,
                             "frac": round((44 * k + 32 * nn + 96 * N) / (acc["qap_ms"] / K * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if acc["qap_ms"] else None},
                "ntt_x6_join": {"algorithmic_bytes": 320 * N,
                                "GBps": round(320 * N / (acc["ntt_ms"] / K * 1e-3) / 1e9, 1) if acc["ntt_ms"] else None,
                                "frac": round(320 * N / (acc["ntt_ms"] / K * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if acc["ntt_ms"] else None,
                                "note": "VALU-bound like the MSM: 317 VALU instructions per butterfly (floor 289), 5.2-6.2 cycles each "
                                        "(profiles/r03_pmc_accumulate.txt)"}},
            "proof_hbm_GBps": round(b_proof / (acc["total_ms"] / K * 1e-3) / 1e9, 2) if acc["total_ms"] else None,
            "roofline": {"bound": "hbm", "kernel": "msm_accumulate_kernel<G1> (avg over its two launches per proof: fused witness group A+B1+C, and H)",
                         "achieved": round(achieved, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 5), "traffic": traffic, "traffic_note": traffic_note,
                         "algorithmic_bytes_per_launch": round(bytes_per_launch),
                         "avg_launch_ms": round(ms_per_launch, 4),
                         "int_roofline": ({
                             "kernel": "H-MSM accumulate launch (uniform 254-bit scalars)",
                             "additions_per_point": -(-255 // info.window_bits[4]),
                             "mads_per_addition": costs["static_v_mad_u64_u32"],
                             "valu_per_addition": costs["static_valu"],
                             "achieved_Tmad_per_s": round(info.n_h * (-(-255 // info.window_bits[4])) * costs["static_v_mad_u64_u32"]
                                                          / max(1e-9, acc["accum"][4] / K * 1e-3) / 1e12, 3),
                             "peak_Tmad_per_s": peak_tmad, "peak_note": peak_note,
                             "note": "v_mad_u64_u32 count = points x window digits (full window precomputation: ceil(255/c) "
                                     "additions per point, c = window bits) x the kernel's static mad count per loop iteration "
                                     "(lib/kernel_costs.json, generated at build time from the compiler's gfx950 assembly)"}
                                          if costs else {"note": "lib/kernel_costs.json missing or generated from other sources: run make"}),
                         "note": "integer-VALU bound: ~2.2k VALU instructions (1.47k v_mad_u64_u32) per mixed addition at ~5 cycles each, 13 (H, c = 20, precomputed windows) to 20 (witness, c = 13) additions per 96-byte point; see DESIGN.md 3.3"},
        }
        if incl_upload is not None:
            out["value_incl_upload"] = incl_upload
        if batch is not None:
            out["batch_throughput"] = batch
            out["batch_verify"] = verify
        if replicas is not None:
            out["replicas_throughput"] = replicas
        if rehearsal is not None:
            out["shard_rehearsal"] = rehearsal
        if world == 1 and not args.no_brackets and args.circuit == "nzcp_live" and args.sha256_blocks == 0:
            prover.close()
            try:
                out["upper_bracket"] = upper_bracket_leg(amd, args, dev, threads, log)
            except AssertionError:
                raise
            except Exception as e:  # noqa: BLE001
                out["upper_bracket"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_plonk and not args.no_cpu:
            prover.close()
            try:
                out["plonk_prover"] = plonk_leg(amd, dev, log)
            except Exception as e:  # noqa: BLE001  (an extra leg must not cost the headline line)
                out["plonk_prover"] = {"error": repr(e)[:300]}
        if world == 1 and not args.no_cpu:
            gpu_bytes = bytes(pr.a) + bytes(pr.b) + bytes(pr.c)
            prover.close()
            out["cpu_baseline"] = cpu_leg(amd, args, proof_obj, pub_list, vkey, zkey, wtns, gpu_bytes, log)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    prover.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

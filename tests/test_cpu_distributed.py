"""N > 1 path on CPU: world_size-2 `gloo` run of the exchanges bench.py does on RCCL (the 768-byte partial-sum
all-gather, and -- split_abc -- the scatter of the A/B/C coset-evaluation slices of the sharded H pipeline).
Each rank owns the point range g16_shard_range gives it, computes its five partial MSM sums (here
with the Python oracle, since there is no GPU), packs the 768-byte partial blob of the C ABI, the
blobs are all-gathered, and every rank assembles the proof with the product's host-only
g16_finish_host.  Result must equal the golden single-GPU proof."""
import json
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, golden_path


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _xyzz_blob(P, g2=False):
    """affine oracle point -> XYZZ Montgomery bytes (zz = zzz = 1), infinity = zeros."""
    import formats as f
    from bn254 import RQ
    if P is None:
        return bytes(256 if g2 else 128)
    one = f.le(RQ)
    if g2:
        return f.g2_to_lem(P) + one + bytes(32) + one + bytes(32)
    return f.g1_to_lem(P) + one + one


def _worker(rank, world, port, name, q, split_abc=False):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import __graft_entry__ as entry
    import formats as f
    import groth16 as g
    from bn254 import G1, G2
    amd = entry.load_package()
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    zkb = open(golden_path(name + ".zkey"), "rb").read()
    meta = json.load(open(golden_path(name + ".json")))
    zk = f.read_zkey(zkb)
    w = f.read_wtns(open(golden_path(name + ".wtns"), "rb").read())["w"]
    n, p, N = zk["nVars"], zk["nPublic"], zk["domainSize"]
    if not split_abc:
        Pv = g.h_scalars(zk, w)          # every rank repeats QAP + 6 NTTs (round-1 scheme)
    else:
        # Sharded H pipeline (g16_shard_begin / g16_shard_end, bench.py step()): rank v mod world evaluates vector v
        # of (A, B, C) on the odd coset, ONE scatter per vector hands rank k its slice [lo_k, hi_k) (equal-size
        # chunks, padded), rank k joins P = A.B - C only there.  Elements travel as 32-byte LE words here (the
        # product ships its 40-byte lazy image; the plan -- owners, ranges, padding -- is the same code).
        from bn254 import R, fr_root
        lohi = [amd.shard_range(N, k, world) for k in range(world)]
        pad = max(h - l for l, h in lohi)
        power = N.bit_length() - 1
        inc = fr_root(power + 1)
        evs = g.build_abc(zk, w)
        mine = []
        for v in range(3):
            owner = amd.shard_vector_owner(v, world)
            recv = torch.zeros(max(1, pad) * 32, dtype=torch.uint8)
            chunks = None
            if rank == owner:
                coef = g.ntt(evs[v], inverse=True)
                sh, t = [], 1
                for x in coef:
                    sh.append(x * t % R)
                    t = t * inc % R
                full = g.ntt(sh) + [0] * pad
                chunks = [torch.frombuffer(bytearray(b"".join(f.le(x) for x in full[l:l + max(1, pad)])), dtype=torch.uint8)
                          for l, _ in lohi]
            dist.scatter(recv, chunks, src=owner)
            raw = recv.numpy().tobytes()
            mine.append([int.from_bytes(raw[i * 32:(i + 1) * 32], "little") for i in range(pad)])
        lo, hi = lohi[rank]
        Pv = [0] * N
        for i in range(lo, hi):
            Pv[i] = (mine[0][i - lo] * mine[1][i - lo] - mine[2][i - lo]) % R

    def part(bases, scalars, total, grp):
        lo, hi = amd.shard_range(total, rank, world)
        return grp.msm(bases[lo:hi], scalars[lo:hi]) if hi > lo else None
    blob = (_xyzz_blob(part(zk["A"], w, n, G1)) + _xyzz_blob(part(zk["B1"], w, n, G1)) +
            _xyzz_blob(part(zk["C"], w[p + 1:], n - p - 1, G1)) + _xyzz_blob(part(zk["H"], Pv, N, G1)) +
            _xyzz_blob(part(zk["B2"], w, n, G2), g2=True))
    assert len(blob) == amd.PARTIAL_BYTES
    mine = torch.frombuffer(bytearray(blob), dtype=torch.uint8)
    gathered = torch.zeros(world * amd.PARTIAL_BYTES, dtype=torch.uint8)
    dist.all_gather_into_tensor(gathered, mine)
    raw = gathered.numpy().tobytes()
    parts = [raw[i * amd.PARTIAL_BYTES:(i + 1) * amd.PARTIAL_BYTES] for i in range(world)]
    proof = amd.finish_host(zkb, parts, f.le(int(meta["r"])), f.le(int(meta["s"])))
    q.put((rank, proof == meta["proof"]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("name,split_abc", [("small", False), ("small", True)])
def test_sharded_exchange_gloo_world2(amd, name, split_abc):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, name, q, split_abc)) for r in range(world)]
    for pr in procs:
        pr.start()
    for pr in procs:
        pr.join(180)
        assert pr.exitcode == 0
    got = sorted(q.get(timeout=10) for _ in range(world))
    assert got == [(0, True), (1, True)]


def test_shard_ranges_partition(amd):
    for total in (0, 1, 7, 513, 1_700_000):
        for count in (1, 2, 3, 8):
            ranges = [amd.shard_range(total, r, count) for r in range(count)]
            assert ranges[0][0] == 0 and ranges[-1][1] == total
            assert all(ranges[i][1] == ranges[i + 1][0] for i in range(count - 1))

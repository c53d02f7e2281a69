"""GPU parity on the edge cases of the path: smallest domains, no public signals, degenerate
witnesses (all zero / all one / all r-1: empty MSMs, one giant 'ones' bucket, carry-heavy digits),
proving keys whose B sections are entirely infinity, and every auto-chosen window size."""
import random

import pytest

import bn254 as b
import formats as f
import groth16 as g
import synth

pytestmark = pytest.mark.gpu


def _prove_both(amd, zkb, w, r, s, **kw):
    zk = f.read_zkey(zkb)
    prover = amd.Prover(zkb, **kw)
    proof, pub = prover.prove(f.write_wtns(w), f.le(r), f.le(s))
    prover.close()
    (A, B, C), opub = g.prove(zk, w, r, s)
    assert proof == f.proof_obj(A, B, C)
    assert pub == [str(x) for x in opub]
    return zk, (A, B, C), opub


@pytest.mark.parametrize("n,p,m", [(2, 0, 1), (3, 1, 1), (5, 0, 2), (9, 3, 4)])
def test_smallest_circuits(amd, n, p, m):
    """domain sizes 2..8, including nPublic = 0 (empty public.json)."""
    rows, w = synth.make(n, p, m, 300 + n)
    assert synth.check_r1cs(rows, w)
    zk, _ = g.setup(n, p, rows, g.trapdoor(301 + n))
    _, proof, pub = _prove_both(amd, f.write_zkey(zk), w, 12345, 67890)
    assert g.verify(zk, pub, proof)
    assert len(pub) == p


@pytest.mark.parametrize("kind", ["zeros", "ones", "minus_one", "r_half"])
def test_degenerate_witnesses(amd, kind):
    """Not satisfying assignments -- parity of the arithmetic only: empty digit lists, every scalar in
    the ones pseudo-window, all-ones bit patterns (every signed digit negative with carries)."""
    n, p, m = 200, 4, 150
    zkb, _, _ = amd.synth_setup(n, p, m, 77)
    val = {"zeros": 0, "ones": 1, "minus_one": b.R - 1, "r_half": (b.R - 1) // 2}[kind]
    w = [1] + [val] * (n - 1)
    _prove_both(amd, zkb, w, 3, 5)


def test_blinding_edge_values(amd):
    zkb, wt, _ = amd.synth_setup(64, 2, 40, 5)
    w = f.read_wtns(wt)["w"]
    for r, s in ((0, 0), (1, b.R - 1), (b.R - 1, b.R - 1)):
        _prove_both(amd, zkb, w, r, s)


def test_all_infinity_b_sections(amd):
    """A circuit whose B polynomials vanish on every signal: B1/B2 MSMs run over zero bases."""
    n, p, m = 40, 2, 20
    rows, w = synth.make(n, p, m, 9)
    rows = [(A, [], C) for (A, _, C) in rows]          # B = 0 everywhere (witness need not satisfy)
    zk, _ = g.setup(n, p, rows, g.trapdoor(10))
    assert all(P is None for P in zk["B1"]) and all(P is None for P in zk["B2"])
    _prove_both(amd, f.write_zkey(zk), w, 11, 13)


@pytest.mark.parametrize("c", [2, 3, 4, 7, 9, 10, 12, 14, 15, 16])
def test_every_window_size(amd, c):
    """window_bits override: narrow top windows (heavy-bucket path) and the widest ones."""
    zkb, wt, _ = amd.synth_setup(500, 5, 400, 21)
    w = f.read_wtns(wt)["w"]
    _prove_both(amd, zkb, w, 7, 9, window_bits=c)


@pytest.mark.parametrize("pf,c", [(2, 0), (3, 5), (4, 16), (16, 13), (200, 4)])
def test_window_precomputation(amd, pf, c):
    """g16_opts.flags = G16_OPT_PRECOMP(pf): base table extended by 2^(c W k) P, scalar windows folded into
    W = ceil(Ws / pf) bucket rows -- same proof bytes."""
    zkb, wt, _ = amd.synth_setup(900, 7, 700, 23)
    w = f.read_wtns(wt)["w"]
    _prove_both(amd, zkb, w, 7, 9, window_bits=c, precomp=pf)


def test_task_len_extremes(amd):
    zkb, wt, _ = amd.synth_setup(700, 513, 100, 22)
    w = f.read_wtns(wt)["w"]
    for tl in (1, 2, 1000):
        _prove_both(amd, zkb, w, 7, 9, task_len=tl)


def test_fft_sizes_one_and_two(amd):
    rng = random.Random(1)
    for n in (1, 2):
        vals = [rng.randrange(b.R) for _ in range(n)]
        mont = b"".join(f.le(v * b.RR % b.R) for v in vals)
        rinv = pow(b.RR, -1, b.R)
        out = amd.fr_fft(mont)
        assert [int.from_bytes(out[i * 32:(i + 1) * 32], "little") * rinv % b.R for i in range(n)] == g.ntt(vals)
        back = amd.fr_fft(out, inverse=True)
        assert back == mont


def test_random_shapes_and_tuning_knobs(amd):
    """24 seeded random combinations of circuit shape, window bits, task length and precomputation factor:
    every proof must equal the big-int oracle's."""
    rng = random.Random(20260103)
    for case in range(24):
        n = rng.choice([7, 33, 120, 257, 600])
        p = rng.choice([0, 1, 5, min(n - 2, 40)])
        m = rng.choice([1, 9, 64, 200, 500])
        c = rng.choice([0, 0, 2, 5, 8, 11, 13, 16])
        tl = rng.choice([0, 0, 1, 3, 16, 500])
        pf = rng.choice([0, 0, 2, 4, 7])
        seed = 5000 + case
        zkb, wt, _ = amd.synth_setup(n, p, m, seed)
        w = f.read_wtns(wt)["w"]
        r, s = rng.randrange(b.R), rng.randrange(b.R)
        _prove_both(amd, zkb, w, r, s, window_bits=c, task_len=tl, precomp=pf)


def test_two_handles_prove_concurrently_from_two_threads(amd):
    """Two resident keys on one GPU, proved at the same time from two host threads (what Node's worker pool
    does with two createProver handles): both results equal their single-threaded proofs."""
    import threading
    zk1, wt1, _ = amd.synth_setup(3000, 513, 2500, 71)
    zk2, wt2, _ = amd.synth_setup(1800, 7, 3000, 72)
    p1, p2 = amd.Prover(zk1), amd.Prover(zk2)
    r, s = f.le(12345), f.le(67890)
    want1, want2 = p1.prove(wt1, r, s), p2.prove(wt2, r, s)
    got, errs = {}, []

    def work(key, prover, wt):
        try:
            for _ in range(6):
                res = prover.prove(wt, r, s)
                if key in got:
                    assert got[key] == res
                got[key] = res
        except Exception as e:  # surfaced below
            errs.append(e)

    ts = [threading.Thread(target=work, args=(1, p1, wt1)), threading.Thread(target=work, args=(2, p2, wt2))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs, errs
    assert got[1] == want1 and got[2] == want2
    p1.close()
    p2.close()


def test_mutated_witness_files_are_refused_or_proved(amd):
    """A .wtns image is untrusted input too: flipped header bytes, truncations and absurd section sizes end in
    G16_E_FORMAT with snarkjs's texts -- or, when only witness VALUES changed, in a proof (of a false statement: the
    prover does not judge the witness) -- never in a fault."""
    import random
    import struct
    from conftest import golden_path
    zk = open(golden_path("small.zkey"), "rb").read()
    w = open(golden_path("small.wtns"), "rb").read()
    prover = amd.Prover(zk)
    rng = random.Random(9)
    outcomes = {"proved": 0, "refused": 0}
    for _ in range(300):
        b = bytearray(w)
        k = rng.randrange(4)
        if k == 0:
            for _j in range(rng.randrange(1, 4)):
                b[rng.randrange(min(len(b), 90))] = rng.randrange(256)
        elif k == 1:
            b = b[:rng.randrange(len(b))]
        elif k == 2:
            i = rng.randrange(len(b) - 4)
            b[i:i + 4] = struct.pack("<I", rng.choice([0, 1, 0xffffffff, 0x7fffffff, rng.randrange(1 << 32)]))
        else:
            i = 12 + rng.randrange(60)
            b[i:i + 8] = struct.pack("<Q", rng.choice([0, 1, len(b), 1 << 40, (1 << 64) - 1]))
        try:
            prover.prove(bytes(b))
            outcomes["proved"] += 1
        except amd.G16Error as e:
            assert e.code == -2, (e.code, str(e))
            outcomes["refused"] += 1
    assert outcomes["refused"] > 100 and outcomes["proved"] > 0
    proof, pub = prover.prove(w)      # the handle is still good
    assert len(pub) == prover.info.n_public
    prover.close()


def test_task_buffer_overflow_is_an_error_not_a_fault(amd, monkeypatch):
    """The bucket-task buffers are sized from the points and the shortest task length; the total a launch really needs is
    known only on the device (it depends on the witness and on the task length picked at run time).  With the buffers
    forced too small (test hook G16_TEST_MAX_TASKS), the device-side capacity check must turn the launch into
    G16_E_STATE -- r02 had a core dump here -- and the handle must stay usable."""
    zkb, wt, _ = amd.synth_setup(3000, 5, 2500, 31)
    monkeypatch.setenv("G16_TEST_MAX_TASKS", "16")
    small = amd.Prover(zkb)
    monkeypatch.delenv("G16_TEST_MAX_TASKS")
    for _ in range(2):      # twice: the failed launch leaves nothing behind
        with pytest.raises(amd.G16Error) as e:
            small.prove(wt, f.le(7), f.le(9))
        assert e.value.code == -5 and "bucket tasks" in str(e.value), e.value
    small.close()
    # the same key and witness on a handle with full-size buffers: the oracle's proof
    _prove_both(amd, zkb, f.read_wtns(wt)["w"], 7, 9)


def test_single_pass_binning_overflow_falls_back(amd, monkeypatch):
    """The dense MSMs' single-pass front end gives every (row, bin) a fixed-capacity region; scalars far from uniform
    overflow it.  With the capacity forced to a few entries (test hook G16_TEST_BIN_CAP) every launch overflows: the
    flag raised on the device makes msm_collect repeat the launch on the two-pass path -- same proof bytes, and the
    handle stays on that path for the next proof."""
    zkb, wt, _ = amd.synth_setup(3000, 5, 2500, 33)
    w = f.read_wtns(wt)["w"]
    monkeypatch.setenv("G16_TEST_BIN_CAP", "3")
    zk = f.read_zkey(zkb)
    prover = amd.Prover(zkb)
    monkeypatch.delenv("G16_TEST_BIN_CAP")
    (A, B, C), opub = g.prove(zk, w, 7, 9)
    for _ in range(2):
        proof, pub = prover.prove(f.write_wtns(w), f.le(7), f.le(9))
        assert proof == f.proof_obj(A, B, C)
    prover.close()


@pytest.mark.parametrize("scan,c", [("1", 0), ("1", 7), ("1", 14), ("1", 17), ("0", 0), ("0", 9)])
def test_both_bucket_reduces_on_every_lane(amd, monkeypatch, scan, c):
    """The product picks the scan-based bucket reduce for dense rows and the per-lane weighting for the witness lanes'
    sparse rows; G16_REDUCE_SCAN forces one of them everywhere.  Both must give the oracle's proof on every kind of row:
    digit rows, the ones row, the salted top window (c = 7, 14), rows of one workgroup, rows cut into several workgroups
    with the FINAL fold on the device (G2: 256-thread workgroups) and with the host fold of triples."""
    zkb, wt, _ = amd.synth_setup(2500, 5, 2000, 41)
    w = f.read_wtns(wt)["w"]
    monkeypatch.setenv("G16_REDUCE_SCAN", scan)
    _prove_both(amd, zkb, w, 7, 9, window_bits=c)


def _repeated_value_witness(n, seed):
    """A witness in the shape that takes the repeated-value rows (MsmGroup::dup_rows): a few dozen full-width values shared
    by 8..300 points each, byte-sized values shared by many, values shared by fewer than kDupMin = 8 points (they stay on the
    digit rows), zeros and ones.  Not a satisfying assignment: parity of the arithmetic only."""
    rnd = random.Random(seed)
    pool = [rnd.randrange(b.R) for _ in range(40)] + [rnd.randrange(256) for _ in range(30)] + [b.R - 1, (b.R - 1) // 2]
    rare = [rnd.randrange(b.R) for _ in range(50)]
    w = [1]
    while len(w) < n:
        k = rnd.random()
        if k < 0.45:
            w.append(rnd.choice(pool))
        elif k < 0.50:
            w.append(rnd.choice(rare))
        elif k < 0.60:
            w.append(rnd.randrange(b.R))
        elif k < 0.80:
            w.append(0)
        else:
            w.append(1)
    return w


@pytest.mark.parametrize("c,chunk,copy", [(13, None, None), (13, "16", None), (11, None, "1"), (16, None, None), (12, "8", None),
                                          (14, "5", None)])
def test_repeated_value_rows_in_every_chunk_width(amd, monkeypatch, c, chunk, copy):
    """Points that share a scalar value enter their section's repeated-value rows once per value; the device multiplies each
    bucket sum by the chunks of its value and the host folds the chunk sums -- merged into the window sums when the chunks are
    as wide as the windows (the default for c <= 16: ONE Horner pass), by a second Horner pass otherwise (G16_DUP_CHUNK; the
    batch pipeline's 16-bit chunks).  Every combination must give the oracle's proof, from g16_prove and from g16_prove_batch,
    also with the row sums copied from a device buffer (G16_ROWS_COPY: r02's path) instead of written to pinned memory."""
    import ctypes as C
    n, p, m = 4000, 4, 2000
    zkb, _, _ = amd.synth_setup(n, p, m, 91)
    w = _repeated_value_witness(n, 92)
    if chunk:
        monkeypatch.setenv("G16_DUP_CHUNK", chunk)
    if copy:
        monkeypatch.setenv("G16_ROWS_COPY", copy)
    zk = f.read_zkey(zkb)
    prover = amd.Prover(zkb, window_bits=c)
    wt = f.write_wtns(w)
    (A, B, Cc), opub = g.prove(zk, w, 7, 9)
    want = f.proof_obj(A, B, Cc)
    proof, pub = prover.prove(wt, f.le(7), f.le(9))
    assert proof == want and pub == [str(x) for x in opub]
    tm = prover.timings()           # read lazily from the events of the proof just made
    assert 0 < tm["qap_ms"] < tm["total_ms"] < 1e3 and tm["msm_ms"][4] > 0, tm
    # the batch pipeline (its own chunk width; more proofs than contexts)
    count = 5
    arr = (C.c_char_p * count)(*([wt] * count))
    lens = (C.c_size_t * count)(*([len(wt)] * count))
    rs = b"".join(f.le(7) + f.le(9) for _ in range(count))
    out = (amd.Proof * count)()
    pubs = C.create_string_buffer(count * p * 32)
    rc = amd.load().g16_prove_batch(prover._h, arr, lens, count, rs, out, pubs)
    assert rc == 0, amd.load().g16_last_error()
    for i in range(count):
        assert amd.proof_to_obj(out[i]) == want, i
    assert prover.timings()["total_ms"] > 0
    # ... and a single proof again on the same handle (the chunk width is chosen per launch)
    proof, _ = prover.prove(wt, f.le(7), f.le(9))
    assert proof == want
    prover.close()

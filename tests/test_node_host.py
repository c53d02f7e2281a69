"""The Node.js host (nzcp-circom_amd/js): snarkjs-shaped `groth16.prove` over the N-API addon.
CPU part: the addon loads, snarkjs error texts surface as thrown Errors, no JS fallback.
GPU part: the CLI twin of `snarkjs groth16 prove` writes byte-identical proof.json/public.json."""
import json
import os
import shutil
import subprocess

import pytest

import formats as f
from conftest import ROOT, golden_path

JS = os.path.join(ROOT, "nzcp-circom_amd", "js")
needs_node = pytest.mark.skipif(shutil.which("node") is None, reason="node not installed")


@pytest.fixture(scope="module")
def addon():
    subprocess.check_call(["make", "-C", os.path.join(JS, "addon")], stdout=subprocess.DEVNULL)
    return os.path.join(JS, "addon", "g16_napi.node")


def run_node(script):
    return subprocess.run(["node", "-e", script], capture_output=True, text=True, timeout=300)


@needs_node
def test_addon_loads_and_reports_snarkjs_errors(addon):
    script = f"""
    const {{ groth16 }} = require({json.dumps(JS)});
    (async () => {{
      const out = [];
      for (const [z, w] of [[{json.dumps(golden_path('tiny.wtns'))}, {json.dumps(golden_path('tiny.wtns'))}],
                            [{{type: "mem", data: new Uint8Array(20)}}, Buffer.alloc(4)]]) {{
        try {{ await groth16.prove(z, w); out.push("ok"); }} catch (e) {{ out.push(e.message); }}
      }}
      console.log(JSON.stringify(out));
    }})();
    """
    r = run_node(script)
    assert r.returncode == 0, r.stderr
    msgs = json.loads(r.stdout)
    assert msgs == ["zkey: Invalid File format", "zkey: Invalid File format"]


@needs_node
@pytest.mark.gpu
@pytest.mark.parametrize("name", ["tiny", "nzcp513"])
def test_cli_writes_byte_identical_json(addon, tmp_path, name):
    meta = json.load(open(golden_path(name + ".json")))
    pj, uj = tmp_path / "proof.json", tmp_path / "public.json"
    r = subprocess.run(["node", os.path.join(JS, "cli.js"), "groth16", "prove", golden_path(name + ".zkey"),
                        golden_path(name + ".wtns"), str(pj), str(uj), "--r", meta["r"], "--s", meta["s"]],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert pj.read_text() == f.js_stringify(meta["proof"])
    assert uj.read_text() == f.js_stringify(meta["public"])


@needs_node
@pytest.mark.gpu
def test_node_verify_and_cli(addon, tmp_path):
    """snarkjs `groth16.verify(vk, publicSignals, proof)` and `snarkjs groth16 verify vk.json public.json proof.json`
    through the Node host (GPU pairing check): the golden proofs verify against the golden verification keys (both
    written by the Python oracle), tampered statements and proofs do not; a resident verifier returns one verdict per
    proof of a batch."""
    meta = json.load(open(golden_path("nzcp513.json")))
    small = json.load(open(golden_path("small.json")))
    script = f"""
    const {{ groth16 }} = require({json.dumps(JS)});
    const meta = {json.dumps({k: meta[k] for k in ('vkey', 'public', 'proof')})};
    const small = {json.dumps({k: small[k] for k in ('vkey', 'public', 'proof')})};
    (async () => {{
      const out = {{}};
      out.good = await groth16.verify(meta.vkey, meta.public, meta.proof);
      const pub2 = meta.public.slice(); pub2[7] = (BigInt(pub2[7]) + 1n).toString();
      out.badPub = await groth16.verify(meta.vkey, pub2, meta.proof);
      const pr2 = JSON.parse(JSON.stringify(meta.proof)); pr2.pi_c = pr2.pi_a;
      out.badProof = await groth16.verify(meta.vkey, meta.public, pr2);
      out.shortPub = await groth16.verify(meta.vkey, meta.public.slice(1), meta.proof);
      out.wrongKey = await groth16.verify(small.vkey, small.public, meta.proof);
      const v = await groth16.createVerifier(small.vkey);
      const items = [];
      for (let i = 0; i < 70; i++) {{
        const ps = small.public.slice();
        if (i % 9 === 4) ps[0] = (BigInt(ps[0]) + BigInt(i)).toString();
        items.push({{ publicSignals: ps, proof: small.proof }});
      }}
      out.batch = await v.verifyBatch(items);
      v.close();
      let err = "none";
      try {{ await groth16.verify({{protocol: "plonk"}}, [], meta.proof); }} catch (e) {{ err = e.message; }}
      out.err = err;
      console.log(JSON.stringify(out));
    }})();
    """
    r = run_node(script)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["good"] is True and out["badPub"] is False and out["badProof"] is False
    assert out["shortPub"] is False and out["wrongKey"] is False
    assert out["batch"] == [i % 9 != 4 for i in range(70)]
    assert "not a groth16 key" in out["err"]
    vk, pj, uj = tmp_path / "verification_key.json", tmp_path / "proof.json", tmp_path / "public.json"
    vk.write_text(json.dumps(meta["vkey"]))
    pj.write_text(f.js_stringify(meta["proof"]))
    uj.write_text(f.js_stringify(meta["public"]))
    r = subprocess.run(["node", os.path.join(JS, "cli.js"), "groth16", "verify", str(vk), str(uj), str(pj)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "snarkJS: OK!" in r.stdout, r.stderr
    bad = list(meta["public"])
    bad[0] = str(int(bad[0]) + 1)
    uj.write_text(f.js_stringify(bad))
    r = subprocess.run(["node", os.path.join(JS, "cli.js"), "groth16", "verify", str(vk), str(uj), str(pj)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "Invalid proof" in r.stderr


@needs_node
def test_zkey_export_verificationkey(tmp_path):
    """`snarkjs zkey export verificationkey` (the second line of the reference's PLONK flow, Makefile:32) from the Node
    host: host-only header reads.  Groth16: the golden key's verification key as the oracle wrote it; PLONK: the
    commitments, k1, k2, X_2 and w of an oracle-written key."""
    import bn254 as b
    import plonk as pk
    import synth
    meta = json.load(open(golden_path("small.json")))
    rows, _w = synth.make(24, 2, 12, 1)
    zk = pk.setup(24, 2, rows, tau=4242)
    zf = tmp_path / "p.zkey"
    zf.write_bytes(pk.write_zkey(zk))
    out = tmp_path / "vk.json"
    r = subprocess.run(["node", os.path.join(JS, "cli.js"), "zkey", "export", "verificationkey", golden_path("small.zkey"), str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = json.loads(out.read_text())
    want = {k: meta["vkey"][k] for k in meta["vkey"] if k != "vk_alphabeta_12"}
    assert got == want and list(got.keys())[:3] == ["protocol", "curve", "nPublic"]
    r = subprocess.run(["node", os.path.join(JS, "cli.js"), "zkey", "export", "verificationkey", str(zf), str(out)],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    got = json.loads(out.read_text())

    def g1(P):
        return [str(P[0]), str(P[1]), "1"]
    assert got["protocol"] == "plonk" and got["nPublic"] == 2 and got["power"] == zk["power"]
    assert (got["k1"], got["k2"]) == (str(zk["k1"]), str(zk["k2"]))
    for k in ("Qm", "Ql", "Qr", "Qo", "Qc", "S1", "S2", "S3"):
        assert got[k] == (g1(zk[k]) if zk[k] is not None else ["0", "1", "0"])
    X2 = zk["X_2"]
    assert got["X_2"] == [[str(X2[0][0]), str(X2[0][1])], [str(X2[1][0]), str(X2[1][1])], ["1", "0"]]
    assert got["w"] == str(b.fr_root(zk["power"]))


@needs_node
@pytest.mark.gpu
def test_node_plonk_setup_from_ptau_files(addon, tmp_path):
    """The reference's own scripted flow (/root/reference/Makefile:31-32) through the Node CLI, on the GPU:
    `plonk setup c.r1cs pot.ptau c.zkey` (file names; powers from a .ptau the oracle wrote for a known tau), then
    `zkey export verificationkey` and `plonk prove` on the result -- the key equals the oracle's byte for byte and the
    proof passes the oracle's verifier with the exported verification key's commitments."""
    import plonk as pk
    import synth
    n, p, m, seed = 60, 5, 40, 3
    _, rows, _ = synth.gen_circuit(n, p, m, seed)
    rows_w, w = synth.make(n, p, m, seed)
    tau = 271828182845
    zk = pk.setup(n, p, rows_w, tau)
    rf, pf, zf, wf = tmp_path / "c.r1cs", tmp_path / "pot.ptau", tmp_path / "c.zkey", tmp_path / "w.wtns"
    rf.write_bytes(f.write_r1cs(n, p, 0, rows))
    pf.write_bytes(pk.write_ptau(zk["power"], tau))
    wf.write_bytes(f.write_wtns(w))
    cli = os.path.join(JS, "cli.js")
    r = subprocess.run(["node", cli, "plonk", "setup", str(rf), str(pf), str(zf), "--lagrange"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert zf.read_bytes() == pk.write_zkey(zk)
    vkf, prf, puf = tmp_path / "vk.json", tmp_path / "proof.json", tmp_path / "public.json"
    assert subprocess.run(["node", cli, "zkey", "export", "verificationkey", str(zf), str(vkf)], capture_output=True, timeout=120).returncode == 0
    assert subprocess.run(["node", cli, "plonk", "prove", str(zf), str(wf), str(prf), str(puf)], capture_output=True, timeout=300).returncode == 0
    vkj = json.loads(vkf.read_text())
    vk = pk.vkey(zk)
    assert vkj["Qm"] == ([str(vk["Qm"][0]), str(vk["Qm"][1]), "1"] if vk["Qm"] else ["0", "1", "0"])
    assert pk.verify(vk, [int(x) for x in json.loads(puf.read_text())], pk.proof_from_obj(json.loads(prf.read_text())))
    # ... and `plonk verify` closes the flow on the GPU: OK for the proof, "Invalid proof" for a changed public signal
    r = subprocess.run(["node", cli, "plonk", "verify", str(vkf), str(puf), str(prf)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "snarkJS: OK!" in r.stdout, r.stderr
    pubs = json.loads(puf.read_text())
    pubs[0] = str(int(pubs[0]) + 1)
    puf.write_text(json.dumps(pubs))
    r = subprocess.run(["node", cli, "plonk", "verify", str(vkf), str(puf), str(prf)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 1 and "Invalid proof" in r.stderr
    r = subprocess.run(["node", cli, "plonk", "setup", str(rf), str(tmp_path / "missing.ptau"), str(zf)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 1 and "cannot open" in r.stderr


@needs_node
@pytest.mark.gpu
def test_node_plonk_prove(addon, tmp_path):
    """snarkjs `plonk.prove(zkey, wtns)` through the Node host: with the oracle's blinding the proof object is the
    oracle's (oracle/plonk.py), with fresh randomness it still passes the oracle's KZG verifier; the CLI twin writes
    proof.json / public.json."""
    import plonk as pk
    import synth
    n, p, m, seed = 60, 5, 40, 3
    rows, w = synth.make(n, p, m, seed)
    zk = pk.setup(n, p, rows, tau=31337)
    zf, wf = tmp_path / "c.zkey", tmp_path / "w.wtns"
    zf.write_bytes(pk.write_zkey(zk))
    wf.write_bytes(f.write_wtns(w))
    rng = synth.Xoshiro(77)
    bl = {i: rng.rand_fr() for i in range(1, 10)}
    script = f"""
    const {{ plonk }} = require({json.dumps(JS)});
    (async () => {{
      const a = await plonk.prove({json.dumps(str(zf))}, {json.dumps(str(wf))}, {{blinding: {json.dumps([str(bl[i]) for i in range(1, 10)])}}});
      const b = await plonk.prove({json.dumps(str(zf))}, {json.dumps(str(wf))});
      let err = "none";
      try {{ await plonk.prove({json.dumps(golden_path('tiny.zkey'))}, {json.dumps(str(wf))}); }} catch (e) {{ err = e.message; }}
      console.log(JSON.stringify({{a, b, err}}));
    }})();
    """
    r = run_node(script)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    exp, exp_pub = pk.prove(zk, w, bl)
    assert out["a"]["proof"] == pk.proof_obj(exp) and out["a"]["publicSignals"] == [str(x) for x in exp_pub]
    assert list(out["a"]["proof"].keys()) == ["A", "B", "C", "Z", "T1", "T2", "T3", "eval_a", "eval_b", "eval_c", "eval_s1",
                                              "eval_s2", "eval_zw", "eval_r", "Wxi", "Wxiw", "protocol", "curve"]
    assert out["b"]["proof"]["A"] != out["a"]["proof"]["A"]
    assert pk.verify(pk.vkey(zk), [int(x) for x in out["b"]["publicSignals"]], pk.proof_from_obj(out["b"]["proof"]))
    assert out["err"] == "zkey file is not plonk"
    pj, uj = tmp_path / "proof.json", tmp_path / "public.json"
    r = subprocess.run(["node", os.path.join(JS, "cli.js"), "plonk", "prove", str(zf), str(wf), str(pj), str(uj)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    assert pk.verify(pk.vkey(zk), [int(x) for x in json.loads(uj.read_text())], pk.proof_from_obj(json.loads(pj.read_text())))


@needs_node
@pytest.mark.gpu
def test_resident_prover_and_random_blinding(addon):
    meta = json.load(open(golden_path("small.json")))
    script = f"""
    const {{ groth16 }} = require({json.dumps(JS)});
    (async () => {{
      const pv = await groth16.createProver({json.dumps(golden_path('small.zkey'))});
      const a = await pv.prove({json.dumps(golden_path('small.wtns'))}, {{r: "{meta['r']}", s: "{meta['s']}"}});
      const [b, c] = await Promise.all([pv.prove({json.dumps(golden_path('small.wtns'))}), pv.prove({json.dumps(golden_path('small.wtns'))})]);
      let bad = "none";
      try {{ await pv.prove({json.dumps(golden_path('tiny.wtns'))}); }} catch (e) {{ bad = e.message; }}
      const w = {json.dumps(golden_path('small.wtns'))};
      const batch = await pv.proveBatch([w, w, w, w, w], {{r: "{meta['r']}", s: "{meta['s']}"}});
      pv.close();
      console.log(JSON.stringify({{a, b, c, bad, batch, info: null}}));
    }})();
    """
    r = run_node(script)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["a"]["proof"] == meta["proof"] and out["a"]["publicSignals"] == meta["public"]
    assert out["b"]["proof"] != out["c"]["proof"] and out["b"]["publicSignals"] == meta["public"]
    assert out["bad"] == "Invalid witness length. Circuit: 150, witness: 24"
    assert len(out["batch"]) == 5
    assert all(x["proof"] == meta["proof"] and x["publicSignals"] == meta["public"] for x in out["batch"])


@needs_node
@pytest.mark.gpu
def test_close_while_proofs_are_in_flight(addon):
    """ADVICE r1 (use-after-free): `p.prove(w).then(..); p.close()` must not delete the native handle under the
    running proof.  JS level: close() waits for requested proofs.  Addon level: destroy() on a handle with a job
    queued or running is deferred until that job retires, and new jobs are refused meanwhile."""
    meta = json.load(open(golden_path("small.json")))
    script = f"""
    const {{ groth16 }} = require({json.dumps(JS)});
    const native = require({json.dumps(addon)});
    const fs = require("fs");
    (async () => {{
      const w = {json.dumps(golden_path('small.wtns'))};
      const pv = await groth16.createProver({json.dumps(golden_path('small.zkey'))});
      const p1 = pv.prove(w, {{r: "{meta['r']}", s: "{meta['s']}"}});
      const p2 = pv.prove(w, {{r: "{meta['r']}", s: "{meta['s']}"}});
      const closed = pv.close();                       // not awaited proofs above
      let after = "none";
      try {{ await pv.prove(w); }} catch (e) {{ after = e.message; }}
      const [a, b] = await Promise.all([p1, p2]);
      await closed;
      // raw addon: destroy right after queueing the job
      const h = await native.create(fs.readFileSync({json.dumps(golden_path('small.zkey'))}), {{}});
      const wb = fs.readFileSync(w);
      const r = Buffer.alloc(32), s = Buffer.alloc(32);
      const job = native.prove(h, wb, r, s);
      native.destroy(h);
      let refused = "none";
      try {{ native.prove(h, wb, r, s); }} catch (e) {{ refused = e.message; }}
      const res = await job;
      console.log(JSON.stringify({{a, b, after, refused, rawlen: res.proof.length}}));
    }})();
    """
    r = run_node(script)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["a"]["proof"] == meta["proof"] and out["b"]["proof"] == meta["proof"]
    assert out["after"] == "prover is closed"
    assert "already destroyed" in out["refused"]
    assert out["rawlen"] == 256


@needs_node
@pytest.mark.gpu
def test_multi_device_prover_from_node(addon):
    """BASELINE config 4 from the product host (VERDICT r1 row e'): createProver(zkey, {devices: [..]}) shards ONE
    proof over the listed GPUs inside the Node process -- here three shards on GPU 0 (a 1-GPU box): MSM point
    ranges + the A/B/C coset evaluations split over the shards, slices moved by device copies, partial sums added
    on the host.  The proof bytes equal the single-GPU proof; snarkjs error texts survive the fan-out."""
    meta = json.load(open(golden_path("nzcp513.json")))
    script = f"""
    const {{ groth16 }} = require({json.dumps(JS)});
    (async () => {{
      const z = {json.dumps(golden_path('nzcp513.zkey'))}, w = {json.dumps(golden_path('nzcp513.wtns'))};
      const pv = await groth16.createProver(z, {{devices: [0, 0, 0]}});
      const a = await pv.prove(w, {{r: "{meta['r']}", s: "{meta['s']}"}});
      const [b, c] = await Promise.all([pv.prove(w), pv.prove(w)]);
      let bad = "none", refused = "none";
      try {{ await pv.prove({json.dumps(golden_path('tiny.wtns'))}); }} catch (e) {{ bad = e.message; }}
      try {{ await groth16.createProver(z, {{shardCount: 2}}); }} catch (e) {{ refused = e.message; }}
      const info = pv.info;
      await pv.close();
      const two = await groth16.prove(z, w, {{devices: [0, 0], r: "{meta['r']}", s: "{meta['s']}"}});
      console.log(JSON.stringify({{a, b, c, bad, refused, info, two}}));
    }})();
    """
    r = run_node(script)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["a"]["proof"] == meta["proof"] and out["a"]["publicSignals"] == meta["public"]
    assert out["two"]["proof"] == meta["proof"] and out["two"]["publicSignals"] == meta["public"]
    assert out["b"]["proof"] != out["c"]["proof"] and out["b"]["publicSignals"] == meta["public"]
    assert "Invalid witness length. Circuit:" in out["bad"]
    assert "devices" in out["refused"]
    assert out["info"]["nPublic"] == 513


@needs_node
def test_nzcp_input_builder_example_pass():
    """pass URI -> ToBeSigned / circuit input / expected public signals, pinned by the
    reference's golden data for the MoH example pass (URI = the test input at
    /root/reference/test/nzcp.js:51, kept as a fixture in tests/golden/example_pass_uri.txt)."""
    uri = open(golden_path("example_pass_uri.txt")).read().strip()
    script = f"""
    const n = require({json.dumps(os.path.join(JS, 'nzcpInput.js'))});
    const uri = {json.dumps(uri)};
    const tbs = n.toBeSigned(uri);
    const inp = n.circuitInput(uri, 314);
    const pub = n.expectedPublicSignals(uri);
    let tooLong = "";
    try {{ n.circuitInput(uri, 300); }} catch (e) {{ tooLong = e.message; }}
    console.log(JSON.stringify({{hex: tbs.toString("hex"), len: inp.toBeSignedLen, nbits: inp.toBeSigned.length,
                                first16: inp.toBeSigned.slice(0, 16), claims: n.claims(uri), npub: pub.length,
                                h1: pub.slice(0, 256).join(""), h2: pub.slice(256, 512).join(""), exp: pub[512],
                                match: n.publicSignalsMatchPass(pub, uri), tooLong}}));
    """
    r = run_node(script)
    assert r.returncode == 0, r.stderr
    o = json.loads(r.stdout)
    assert o["len"] == 314 and o["nbits"] == 314 * 8
    assert o["hex"].startswith("846a5369676e6174757265314aa204456b65792d3101264059011fa501781e6469643a7765623a6e7a6370")
    import hashlib
    assert hashlib.sha256(bytes.fromhex(o["hex"])).hexdigest() == "271ce33d671a2d3b816d788135f4343e14bc66802f8cd841faac939e8c11f3ee"
    assert o["first16"] == [1, 0, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 1, 0, 1, 0]          # 0x84 0x6a, MSB first
    assert (o["claims"]["givenName"], o["claims"]["familyName"], o["claims"]["dob"]) == ("Jack", "Sparrow", "1960-04-16")
    assert o["npub"] == 513 and o["exp"] == "1951416330" and o["match"] is True
    assert "%064x" % int(o["h1"], 2) == "5fb355822221720ea4ce6734e5a09e459d452574a19310c0cea7c141f43a3dab"
    assert "%064x" % int(o["h2"], 2) == "271ce33d671a2d3b816d788135f4343e14bc66802f8cd841faac939e8c11f3ee"
    assert "circuit maximum is 300" in o["tooLong"]


@needs_node
@pytest.mark.gpu
def test_node_prove_of_example_pass_matches_the_pass(addon, amd, tmp_path):
    """End to end in the reference's host language: `groth16.prove` (N-API -> C ABI -> HIP) on the fixed-layout NZCP
    interface circuit over the MoH example pass, then nzcpInput.publicSignalsMatchPass(publicSignals, uri) -- the
    check /root/reference/test/nzcp.js:41-49 makes on the circuit's outputs."""
    from test_cpu_sha256_circuit import EXAMPLE_EXP_OFF, EXAMPLE_SEGS, example_public_signals, example_to_be_signed
    out = amd.nzcp_fixed_layout_setup(example_to_be_signed(), EXAMPLE_SEGS, EXAMPLE_EXP_OFF, 31337)
    zk, wt = tmp_path / "nzcp_fixed.zkey", tmp_path / "nzcp_fixed.wtns"
    zk.write_bytes(out["zkey"])
    wt.write_bytes(out["wtns"])
    uri = open(golden_path("example_pass_uri.txt")).read().strip()
    script = f"""
    const {{ groth16 }} = require({json.dumps(JS)});
    const nz = require({json.dumps(os.path.join(JS, "nzcpInput.js"))});
    (async () => {{
      const {{ proof, publicSignals }} = await groth16.prove({json.dumps(str(zk))}, {json.dumps(str(wt))});
      console.log(JSON.stringify({{ok: nz.publicSignalsMatchPass(publicSignals, {json.dumps(uri)}), n: publicSignals.length,
                                  last: publicSignals[512], protocol: proof.protocol, curve: proof.curve}}));
    }})();
    """
    r = run_node(script)
    assert r.returncode == 0, r.stderr
    o = json.loads(r.stdout)
    assert o == {"ok": True, "n": 513, "last": str(example_public_signals()[512]), "protocol": "groth16", "curve": "bn128"}


@needs_node
def test_synthetic_pass_generator_reproduces_the_example_and_the_live_offsets():
    """syntheticPass: with the example's cti and signature it must reproduce the MoH example URI
    byte for byte (/root/reference/test/nzcp.js:51); in live format the claims map / exp / vc / credentialSubject
    sit at 30 / 72 / 80 / 250 of ToBeSigned (/root/reference/circuits/nzcptpl.circom:438,
    /root/reference/test/nzcp.js:103,158,230), the example's at 27 / 68 / 76 / 246."""
    uri = open(golden_path("example_pass_uri.txt")).read().strip()
    script = f"""
    const n = require({json.dumps(os.path.join(JS, "nzcpInput.js"))});
    const ex = {json.dumps(uri)};
    const same = n.syntheticPass({{format: "example", cti: Buffer.from("60a4f54d4e304332be33ad78b1eafa4b", "hex"),
                                  signature: n.parsePassURI(ex).signature}});
    const live = n.syntheticPass({{givenName: "Aroha", familyName: "Ngata-Smith", dob: "1987-11-02", exp: 1767225600}});
    console.log(JSON.stringify({{same: same === ex, exl: n.fixedLayout(ex), livel: n.fixedLayout(live), c: n.claims(live),
                                pub: n.expectedPublicSignals(live).length, again: n.syntheticPass({{givenName: "Aroha",
                                familyName: "Ngata-Smith", dob: "1987-11-02", exp: 1767225600}}) === live}}));
    """
    r = run_node(script)
    assert r.returncode == 0, r.stderr
    o = json.loads(r.stdout)
    assert o["same"] and o["again"] and o["pub"] == 513
    ex, lv = o["exl"], o["livel"]
    assert (ex["claimsAt"], ex["expAt"], ex["vcAt"], ex["credSubjAt"], ex["toBeSignedLen"]) == (27, 68, 76, 246, 314)
    assert ex["segs"] == [[258, 4], [274, 7], [286, 10]]
    assert (lv["claimsAt"], lv["expAt"], lv["vcAt"], lv["credSubjAt"]) == (30, 72, 80, 250)
    assert lv["toBeSignedLen"] <= 355
    assert o["c"] == {"exp": 1767225600, "nbf": 1635883530, "iss": "did:web:nzcp.identity.health.nz",
                      "givenName": "Aroha", "familyName": "Ngata-Smith", "dob": "1987-11-02"}


@needs_node
@pytest.mark.gpu
def test_node_prove_of_a_synthetic_live_format_pass(addon, amd, tmp_path):
    """config 2's public interface on a live-FORMAT pass: generate the pass, derive the circuit constants with
    fixedLayout, key the fixed-layout circuit, prove through the N-API addon, check public.json against the pass."""
    gen = f"""
    const n = require({json.dumps(os.path.join(JS, "nzcpInput.js"))});
    const uri = n.syntheticPass({{givenName: "Hemi", familyName: "Walker", dob: "2001-02-28", exp: 1798761600}});
    console.log(JSON.stringify({{uri, tbs: n.toBeSigned(uri).toString("hex"), lay: n.fixedLayout(uri)}}));
    """
    r = run_node(gen)
    assert r.returncode == 0, r.stderr
    g0 = json.loads(r.stdout)
    tbs, lay = bytes.fromhex(g0["tbs"]), g0["lay"]
    out = amd.nzcp_fixed_layout_setup(tbs, [tuple(x) for x in lay["segs"]], lay["expOff"], 20260101)
    zk, wt = tmp_path / "live.zkey", tmp_path / "live.wtns"
    zk.write_bytes(out["zkey"])
    wt.write_bytes(out["wtns"])
    script = f"""
    const {{ groth16 }} = require({json.dumps(JS)});
    const nz = require({json.dumps(os.path.join(JS, "nzcpInput.js"))});
    (async () => {{
      const {{ publicSignals }} = await groth16.prove({json.dumps(str(zk))}, {json.dumps(str(wt))});
      console.log(JSON.stringify({{ok: nz.publicSignalsMatchPass(publicSignals, {json.dumps(g0["uri"])}), exp: publicSignals[512]}}));
    }})();
    """
    r = run_node(script)
    assert r.returncode == 0, r.stderr
    assert json.loads(r.stdout) == {"ok": True, "exp": "1798761600"}

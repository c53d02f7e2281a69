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
      pv.close();
      console.log(JSON.stringify({{a, b, c, bad, info: null}}));
    }})();
    """
    r = run_node(script)
    assert r.returncode == 0, r.stderr
    out = json.loads(r.stdout)
    assert out["a"]["proof"] == meta["proof"] and out["a"]["publicSignals"] == meta["public"]
    assert out["b"]["proof"] != out["c"]["proof"] and out["b"]["publicSignals"] == meta["public"]
    assert out["bad"] == "Invalid witness length. Circuit: 150, witness: 24"

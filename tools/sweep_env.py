#!/usr/bin/env python3
"""Environment-knob sweep on the GPU box: one bench.py run per setting, prints ms/proof (single and batch).
usage: python tools/sweep_env.py "G16_SEG_LEN=8,4,8" "G16_WINDOW_BITS=15,0 G16_ACC_WAVES=03" ...
(an empty string = defaults).  Output also appended to gpurun_out/sweep.txt."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    settings = sys.argv[1:] or [""]
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    extra = os.environ.get("SWEEP_BENCH_ARGS", "--steps 40 --warmup 5 --no-cpu --batch-streams 0").split()
    with open(os.path.join(ROOT, "gpurun_out", "sweep.txt"), "a") as log:
        for st in settings:
            env = dict(os.environ)
            more = []
            for kv in st.split():
                if kv.startswith("--"):       # a bench.py argument, e.g. --circuit=synthetic
                    more.append(kv)
                    continue
                k, v = kv.split("=", 1)
                env[k] = v
            try:
                out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra + more, env=env,
                                     capture_output=True, text=True, timeout=280)
                line = out.stdout.strip().splitlines()[-1]
                d = json.loads(line)
                bt = d.get("batch_throughput") or {}
                ph = d["phases_ms"]
                p50, mn = d.get("ms_per_step_p50_min", [0, 0])
                msg = (f"{st or 'default':50s} single {d['ms_per_step']:7.3f} (p50 {p50:6.3f} min {mn:6.3f}) ms  batch {bt.get('ms_per_proof', 0):7.3f} ms  "
                       f"qap {ph['qap']:.2f} ntt {ph['ntt_x6_join']:.2f} msmW/G2/H {ph['msm_A_B1_B2_C_H'][0]:.2f}/"
                       f"{ph['msm_A_B1_B2_C_H'][2]:.2f}/{ph['msm_A_B1_B2_C_H'][4]:.2f}")
            except Exception as e:  # noqa: BLE001
                msg = f"{st or 'default':50s} FAILED {e!r} {out.stderr[-300:] if 'out' in dir() else ''}"
            print(msg, flush=True)
            log.write(msg + "\n")
            log.flush()


if __name__ == "__main__":
    main()

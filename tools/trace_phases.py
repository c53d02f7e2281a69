#!/usr/bin/env python3
"""Average the `[g16 dev]` timeline lines that G16_TRACE_HOST=1 prints (one block per proof) into phase
durations.  usage: G16_SERIAL_MSM=1 G16_TRACE_HOST=1 python bench.py ... 2> err.txt; python tools/trace_phases.py err.txt [skip]
(skip = number of leading proofs to ignore: warm-up).  In serial mode the durations are standalone stage times."""
import re
import sys


def main():
    path = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    blocks, cur = [], {}
    tails, tcur = [], {}
    for line in open(path):
        if line.startswith("[g16 tail]"):   # the host's second half of the proof: "<what> at <ms> ms"
            m = re.match(r"\[g16 tail\] (.*) at ([\d.]+) ms", line)
            if m:
                tcur[m.group(1)] = float(m.group(2))
                if m.group(1) == "pi_c assembled":
                    tails.append(tcur)
                    tcur = {}
            continue
        if not line.startswith("[g16 dev]"):
            continue
        body = line[len("[g16 dev]"):].strip()
        if body.startswith("qap"):
            if cur:
                blocks.append(cur)
            cur = {}
            m = re.match(r"qap 0\.\.([\d.]+)\s+ntt\+join \.\.([\d.]+)\s+total ([\d.]+)", body)
            cur["qap"], cur["ntt_end"], cur["total"] = map(float, m.groups())
        else:
            tag = body.split()[0]
            nums = [float(x) for x in re.findall(r"(?<![A-Za-z\d])(\d+\.\d+)", body)]
            cur[tag] = nums
    if cur:
        blocks.append(cur)
    blocks = blocks[skip:]
    if not blocks:
        print("no blocks")
        return
    n = len(blocks)

    def avg(f):
        return sum(f(b) for b in blocks) / n
    print(f"{n} proofs: qap {avg(lambda b: b['qap']):.3f}  ntt+join end {avg(lambda b: b['ntt_end']):.3f}  total {avg(lambda b: b['total']):.3f}")
    for tag in ("W", "H"):
        if tag not in blocks[0]:
            continue
        # start pass0 binscan pass1 binsort | queue acc0 acc1 combine reduce end streamend
        names = ["pass0", "binscan", "pass1", "binsort", "queue", "acc_start", "accumulate", "combine", "reduce", "tree+copy"]
        v = [avg(lambda b, i=i: b[tag][i]) for i in range(12)]
        d = [v[1] - v[0], v[2] - v[1], v[3] - v[2], v[4] - v[3], v[5] - v[4], v[6] - v[5], v[7] - v[6], v[8] - v[7], v[9] - v[8], v[10] - v[9]]
        print(f"  {tag}: start {v[0]:.3f} " + " ".join(f"{k} {x:.3f}" for k, x in zip(names, d)) + f" | end {v[10]:.3f}")
    if "W2" in blocks[0]:
        v = [avg(lambda b, i=i: b["W2"][i]) for i in range(6)]
        print(f"  W2 (G2 lane): queue@{v[0]:.3f} accumulate {v[2] - v[1]:.3f} combine {v[3] - v[2]:.3f} reduce {v[4] - v[3]:.3f} tree+copy {v[5] - v[4]:.3f} | end {v[5]:.3f}")
    tails = tails[skip:]
    if tails:
        keys = [k for k in tails[0] if all(k in t for t in tails)]
        print("  host (ms since the launch began): " + "; ".join(f"{k} {sum(t[k] for t in tails) / len(tails):.3f}" for k in keys))


if __name__ == "__main__":
    main()

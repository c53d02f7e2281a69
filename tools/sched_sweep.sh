#!/bin/bash
# scheduling sweep (round 1): accumulate gating / persistent-grid size; prints proofs/s, ms, device timeline
run() {
  echo "== GATE=$1 ACC_WAVES=$2"
  G16_TRACE_HOST=1 G16_GATE=$1 G16_ACC_WAVES=$2 timeout -k 10 200 python bench.py --steps 8 --warmup 2 --no-cpu --batch-streams ${3:-0} 2> gpurun_out/sched.err | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],(d.get('batch_throughput') or {}).get('proofs_per_sec'))"
  grep "g16 dev" gpurun_out/sched.err | tail -6
}
for cfg in "$@"; do run ${cfg%%:*} ${cfg##*:}; done

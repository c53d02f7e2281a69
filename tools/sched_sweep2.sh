#!/bin/bash
# one line per configuration: env assignments given as arguments "K=V,K=V"
for cfg in "$@"; do
  envs=$(echo "$cfg" | tr ',' ' ')
  r=$(env $envs timeout -k 10 200 python bench.py --steps 10 --warmup 2 --no-cpu 2>/dev/null | python3 -c "import json,sys;d=json.loads(sys.stdin.read());print(d['value'],d['ms_per_step'],d['phases_ms']['msm_A_B1_B2_C_H'],(d.get('batch_throughput') or {}).get('proofs_per_sec'))")
  echo "$cfg => $r"
done

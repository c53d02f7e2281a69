#!/bin/bash
# A/B harness: untraced single-proof latency (median of the per-step times) of the default workload under env knobs
#   usage: bash tools/r03_ab.sh OUTDIR [--circuit synthetic] -- name1 ENV=.. ENV=.. -- name2 ...
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
EXTRA=""
while [ "$1" != "--" ] && [ $# -gt 0 ]; do EXTRA="$EXTRA $1"; shift; done
shift
while [ $# -gt 0 ]; do
  name=$1; shift
  envs=""
  while [ "$1" != "--" ] && [ $# -gt 0 ]; do envs="$envs $1"; shift; done
  shift
  env $envs timeout -k 10 200 python bench.py --steps 30 --warmup 5 --no-cpu --no-plonk --no-brackets --batch-streams 0 $EXTRA > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -3 $OUT/$name.err; continue; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json"))
print("%-14s ms/step %.3f  p50 %.3f min %.3f  phases %s" % ("$name", d["ms_per_step"], d["ms_per_step_p50_min"][0], d["ms_per_step_p50_min"][1], d["phases_ms"]))
PY
done

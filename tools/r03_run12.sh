#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03l
rm -rf $OUT && mkdir -p $OUT
for v in "c13 X=1" "c12 G16_WINDOW_BITS=12,0" "c11 G16_WINDOW_BITS=11,0" "c14 G16_WINDOW_BITS=14,0" "c13b X=2" "c12b G16_WINDOW_BITS=12,0"; do
  set -- $v
  env $2 timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-proofs 384 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
d=json.load(open("$OUT/$1.json")); print("$1 single %.3f (p50 %.3f) batch %.1f" % (d["ms_per_step"], d["ms_per_step_p50_min"][0], d["batch_throughput"]["proofs_per_sec"]))
PY
done
for v in "syn_c13 X=1" "syn_c12 G16_WINDOW_BITS=12,0" "syn_c14 G16_WINDOW_BITS=14,0"; do
  set -- $v
  env $2 timeout -k 10 300 python bench.py --circuit synthetic --steps 20 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-proofs 96 > $OUT/$1.json 2> $OUT/$1.err
  python - <<PY
import json
d=json.load(open("$OUT/$1.json")); print("$1 single %.3f (p50 %.3f) batch %.1f" % (d["ms_per_step"], d["ms_per_step_p50_min"][0], d["batch_throughput"]["proofs_per_sec"]))
PY
done

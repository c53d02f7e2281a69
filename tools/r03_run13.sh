#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03m
rm -rf $OUT && mkdir -p $OUT
bash tools/r03_ab.sh r03m -- base X=1 -- wide G16_DIRECT_THREADS=512 -- per2048 G16_BIN_PER=2048 -- wide2048 G16_DIRECT_THREADS=512 G16_BIN_PER=2048 -- base2 X=2 -- wideb G16_DIRECT_THREADS=512
for v in "base X=1" "wide G16_DIRECT_THREADS=512" "per2048 G16_BIN_PER=2048" "wide2048 G16_DIRECT_THREADS=512,G16_BIN_PER=2048"; do
  set -- $v
  envs=$(echo $2 | tr ',' ' ')
  env $envs G16_SERIAL_MSM=1 G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_$1.err
  echo "# serial $1"; python tools/trace_phases.py $OUT/trace_$1.err 3 | grep "H:"
done
for v in "h20 X=1" "h19 G16_WINDOW_BITS=0,19"; do
  set -- $v
  env $2 timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-proofs 512 > $OUT/batch_$1.json 2> $OUT/batch_$1.err
  python - <<PY
import json
d=json.load(open("$OUT/batch_$1.json")); print("batch $1", d["batch_throughput"]["proofs_per_sec"], d["ms_per_step"])
PY
done

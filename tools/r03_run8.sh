#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/r03_ab.sh r03h -- L8 X=1 -- L16 G16_LIB_NAME=libg16hip_ab.so -- L8b X=2 -- L16b G16_LIB_NAME=libg16hip_ab.so
bash tools/r03_ab.sh r03h --circuit synthetic -- syn_L8 X=1 -- syn_L16 G16_LIB_NAME=libg16hip_ab.so
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03h
for v in L8 L16; do
  [ $v = L16 ] && export G16_LIB_NAME=libg16hip_ab.so
  timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-proofs 512 > $OUT/batch_$v.json 2> $OUT/batch_$v.err
  python - <<PY
import json
d=json.load(open("$OUT/batch_$v.json")); print("batch $v", d["batch_throughput"]["proofs_per_sec"], d["ms_per_step"])
PY
done
unset G16_LIB_NAME
timeout -k 10 300 python -m pytest tests/test_gpu_prove.py tests/test_gpu_edges.py -x -q 2>&1 | tail -3
for m in conc serial; do
  [ $m = serial ] && export G16_SERIAL_MSM=1
  G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_$m.err
  echo "# $m"; python tools/trace_phases.py $OUT/trace_$m.err 3
done

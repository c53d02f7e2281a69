#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03k
rm -rf $OUT && mkdir -p $OUT
export G16_SERIAL_MSM=1
G16_TRACE_HOST=1 timeout -k 10 300 python bench.py --sha256-blocks 163 --steps 6 --warmup 2 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_serial_2p22.err
echo "# serial sha256x163 (2^22)"; python tools/trace_phases.py $OUT/trace_serial_2p22.err 2
unset G16_SERIAL_MSM
timeout -k 10 300 python bench.py --sha256-blocks 163 --steps 10 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-streams 0 > $OUT/sha.json 2> $OUT/sha.err
python - <<PY
import json
d=json.load(open("$OUT/sha.json")); print("sha256x163", d["ms_per_step"], d["phases_ms"])
PY
timeout -k 10 400 python tools/plonk_bench.py > $OUT/plonk.json 2> $OUT/plonk.err; tail -c 1200 $OUT/plonk.json
timeout -k 10 600 python -m pytest tests/test_gpu_plonk.py "tests/test_gpu_fullsize.py::test_config5_stress_2_22" -x -q 2>&1 | tail -3

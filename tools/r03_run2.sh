#!/bin/bash
# round-3 GPU pass 2: the tests touched by the changes, timelines, and a few knob sweeps
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03b
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_prove.py tests/test_gpu_layers.py "tests/test_gpu_fullsize.py::test_config4_eight_shards_on_the_live_circuit" "tests/test_gpu_fullsize.py::test_real_nzcp_circuit_with_in_circuit_cbor_search" -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
run() {  # name, env...
  name=$1; shift
  env "$@" G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --steps 10 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-proofs 192 > $OUT/$name.json 2> $OUT/$name.err || { echo "$name FAILED"; tail -5 $OUT/$name.err; return 1; }
  python - <<PY
import json
d=json.load(open("$OUT/$name.json"))
print("$name", d["ms_per_step"], d["ms_per_step_p50_min"], "batch", d["batch_throughput"]["proofs_per_sec"])
PY
  echo "## $name" >> $OUT/timelines.txt; python tools/trace_phases.py $OUT/$name.err 3 >> $OUT/timelines.txt
}
run base X=1 || exit 1
run serial G16_SERIAL_MSM=1 || exit 1
run hcalls G16_TAIL_CALLS=1 || exit 1
run nocalls G16_TAIL_CALLS=0 || exit 1
run seg844 G16_SEG_LEN=8,4,4 || exit 1
run seg488 G16_SEG_LEN=4,8,8 || exit 1
run seg8816 G16_SEG_LEN=8,8,16 || exit 1
run tl64 G16_TASK_LEN=0,64 || exit 1
cat $OUT/timelines.txt
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-plonk --batch-proofs 64 > $OUT/bench_full.json 2> $OUT/bench_full.err; echo "bench_full rc=$?"; tail -3 $OUT/bench_full.err
python - <<PY
import json
d=json.load(open("$OUT/bench_full.json"))
print(json.dumps(d.get("shard_rehearsal"))[:1500]); print(json.dumps(d.get("upper_bracket"))[:1500])
PY

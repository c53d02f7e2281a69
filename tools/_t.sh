cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03z
timeout -k 10 400 python -X faulthandler -m pytest tests/test_gpu_prove.py tests/test_gpu_edges.py tests/test_gpu_layers.py -x -q --timeout 100 -p no:cacheprovider > gpurun_out/r03z/diag.log 2>&1; tail -1 gpurun_out/r03z/diag.log
G16_SERIAL_MSM=1 G16_TRACE_HOST=1 timeout -k 10 120 python bench.py --steps 10 --warmup 2 --no-cpu --no-plonk --no-brackets --batch-proofs 0 --batch-streams 0 > gpurun_out/r03z/t.json 2> gpurun_out/r03z/t.err; python tools/trace_phases.py gpurun_out/r03z/t.err 3 | grep " H:"
for v in "a X=1" "a2 X=2" "a3 X=3"; do
  set -- $v
  env $2 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu --no-plonk --batch-proofs 512 > gpurun_out/r03z/$1.json 2> gpurun_out/r03z/$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03z/$1.json")); r=d["shard_rehearsal"]
print("$1 single %.3f batch %.1f upper %.3f" % (d["ms_per_step"], d["batch_throughput"]["proofs_per_sec"], d["upper_bracket"]["ms_per_proof"]), {k:(v["critical_path_ms_excl_exchange"]) for k,v in r.items() if k!="note"}, d.get("phases_ms")["device_total"])
PY
done

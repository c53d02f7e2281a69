cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03z
timeout -k 10 500 python -X faulthandler -m pytest tests/test_gpu_prove.py tests/test_gpu_edges.py "tests/test_gpu_fullsize.py::test_config3_batch_throughput_mode" "tests/test_gpu_fullsize.py::test_config3_batch_nominal" -x -q --timeout 200 -p no:cacheprovider > gpurun_out/r03z/diag.log 2>&1; tail -1 gpurun_out/r03z/diag.log
for v in "a X=1" "a2 X=2" "a3 X=3"; do
  set -- $v
  env $2 timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu --no-plonk --batch-proofs 768 > gpurun_out/r03z/$1.json 2> gpurun_out/r03z/$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03z/$1.json")); r=d["shard_rehearsal"]
print("$1 single %.3f batch %.1f upper %.3f upper batch %.1f" % (d["ms_per_step"], d["batch_throughput"]["proofs_per_sec"], d["upper_bracket"]["ms_per_proof"], d["upper_bracket"]["batch"]["proofs_per_sec"]), {k:(v["critical_path_ms_excl_exchange"]) for k,v in r.items() if k!="note"}, d.get("phases_ms")["device_total"])
PY
done

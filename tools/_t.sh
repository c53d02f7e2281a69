cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r03g
for v in "base X=1" "g11 G16_GATE=11" "g01 G16_GATE=01" "g10 G16_GATE=10" "g22 G16_GATE=22" "g02 G16_GATE=02" "g12 G16_GATE=12" "base2 X=1" "g11p G16_GATE=11,G16_CHAIN_PRIO=1"; do
  set -- $v
  envs=$(echo $2 | tr ',' ' ')
  env $envs timeout -k 10 200 python bench.py --steps 30 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-proofs 256 > gpurun_out/r03g/$1.json 2> gpurun_out/r03g/$1.err
  python - <<PY
import json
d=json.load(open("gpurun_out/r03g/$1.json"))
print("$1 single %.3f batch %.1f" % (d["ms_per_step"], d["batch_throughput"]["proofs_per_sec"]), d.get("phases_ms"))
PY
done

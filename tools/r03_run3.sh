#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03c
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_prove.py tests/test_gpu_layers.py -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
bash tools/r03_ab.sh r03c -- base X=1 -- nocalls G16_TAIL_CALLS=0 -- allcalls G16_TAIL_CALLS=1 -- nodirect G16_NO_DIRECT_BIN=1 -- noworder G16_NO_WINDOW_ORDER=1 -- uniform G16_UNIFORM_WINDOWS=1 G16_NO_DIRECT_BIN=1 -- wg256 G16_REDUCE_WG=256 -- seg848 G16_SEG_LEN=8,4,8 -- seg884 G16_SEG_LEN=8,8,4 -- nooffs G16_NO_REDUCE_OFFSETS=1 -- base2 X=2
# device timelines without the batch leg
for m in conc serial; do
  [ $m = serial ] && export G16_SERIAL_MSM=1
  G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_$m.err
  echo "# $m"; python tools/trace_phases.py $OUT/trace_$m.err 3
done
G16_TAIL_CALLS=0 G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_serial_nocalls.err
echo "# serial nocalls"; python tools/trace_phases.py $OUT/trace_serial_nocalls.err 3
# per-kernel durations, serial mode (standalone), calls vs no calls
cd /tmp && export TMPDIR=/tmp
for v in calls nocalls; do
  [ $v = nocalls ] && export G16_TAIL_CALLS=0
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_$v -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/prof_$v.err
  f=$(find $OUT/prof_$v -name "*kernel_stats.csv" | head -1)
  echo "## $v"; python3 - <<PY
import csv
rows=list(csv.DictReader(open("$f")))
for r in rows:
    n=r["Name"]
    if "msm_" in n or "ntt" in n or "qap" in n:
        short=n.split("(")[0].replace("void g16::","").replace("g16::","")[:70]
        print("%-72s calls %4s avg_us %9.1f" % (short, r["Calls"], float(r["AverageNs"])/1e3))
PY
done

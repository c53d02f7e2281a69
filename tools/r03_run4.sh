#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03d
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_prove.py tests/test_gpu_layers.py -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
bash tools/r03_ab.sh r03d -- base X=1 -- scan1 G16_REDUCE_SCAN=1 -- scan0 G16_REDUCE_SCAN=0 -- nodirect G16_NO_DIRECT_BIN=1 -- per2048 G16_BIN_PER=2048 -- per1024 G16_BIN_PER=1024 -- uniform G16_UNIFORM_WINDOWS=1 G16_NO_DIRECT_BIN=1 -- r02like G16_UNIFORM_WINDOWS=1 G16_NO_DIRECT_BIN=1 G16_REDUCE_SCAN=0 -- worder G16_WINDOW_ORDER=1 -- base2 X=2 -- base3 X=3
bash tools/r03_ab.sh r03d --circuit synthetic -- syn_base X=1 -- syn_r02like G16_UNIFORM_WINDOWS=1 G16_NO_DIRECT_BIN=1 G16_REDUCE_SCAN=0 -- syn_nodirect G16_NO_DIRECT_BIN=1 -- syn_uniform G16_UNIFORM_WINDOWS=1 G16_NO_DIRECT_BIN=1
for m in conc serial; do
  [ $m = serial ] && export G16_SERIAL_MSM=1
  G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_$m.err
  echo "# $m"; python tools/trace_phases.py $OUT/trace_$m.err 3
done

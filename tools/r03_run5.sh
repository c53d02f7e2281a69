#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03e
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_gpu_edges.py tests/test_gpu_layers.py -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -3 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
bash tools/r03_ab.sh r03e -- base X=1 -- tile10 G16_NTT_TILE_LOG=10 -- tile11 G16_NTT_TILE_LOG=11 -- tile10tb3 G16_NTT_TILE_LOG=10 G16_NTT_MIN_TB=3 -- h19 G16_WINDOW_BITS=0,19 -- h21 G16_WINDOW_BITS=0,21 -- h19u G16_WINDOW_BITS=0,19 G16_UNIFORM_WINDOWS=1 G16_NO_DIRECT_BIN=1 -- base2 X=2
bash tools/r03_ab.sh r03e --circuit synthetic -- syn_base X=1 -- syn_tile10 G16_NTT_TILE_LOG=10 -- syn_tile11 G16_NTT_TILE_LOG=11
# standalone NTT chain at 2^20 / 2^21 / 2^22 (serial mode: one stream)
export G16_SERIAL_MSM=1
for c in "nzcp_live" "synthetic" ; do
  G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --circuit $c --steps 8 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_serial_$c.err
  echo "# serial $c"; python tools/trace_phases.py $OUT/trace_serial_$c.err 3
done
G16_TRACE_HOST=1 timeout -k 10 300 python bench.py --sha256-blocks 163 --steps 6 --warmup 2 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_serial_2p22.err
echo "# serial sha256x163 (2^22)"; python tools/trace_phases.py $OUT/trace_serial_2p22.err 2
# PMC: instruction cache + issue counters of the tails, serial mode
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $OUT/pmc_ic -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/pmc_ic.err
python3 $GRAFT_REPO_ROOT/tools/pmc_dump.py $OUT/pmc_ic msm_ > $OUT/pmc_ic.txt; cat $OUT/pmc_ic.txt
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_WAVES SQ_IFETCH -d $OUT/pmc_sq -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/pmc_sq.err
python3 $GRAFT_REPO_ROOT/tools/pmc_dump.py $OUT/pmc_sq > $OUT/pmc_sq.txt; grep -E "reduce|combine|dup_bits|accumulate|fold|ntt|wave_reduce" $OUT/pmc_sq.txt

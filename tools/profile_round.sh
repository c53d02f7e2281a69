#!/bin/bash
# Round profile on the GPU box (run through gpurun from the repo root): default bench line, rocprofv3 kernel
# stats of the same command, and three PMC passes in serial-MSM mode (one MSM at a time, so a launch's counters
# are its own).  Outputs under gpurun_out/final/; tools/pmc_summary.py turns them into profiles/*.
set -e
OUT=$GRAFT_REPO_ROOT/gpurun_out/final
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
# the measured integer-issue peak bench.py prices the accumulate kernel against
timeout -k 10 120 nzcp-circom_amd/lib/microbench_bench > $OUT/microbench.txt 2>&1 && python tools/microbench_summary.py $OUT/microbench.txt $OUT/microbench_int_rates.json
mkdir -p profiles && cp $OUT/microbench_int_rates.json profiles/r03_microbench_int_rates.json
timeout -k 10 900 python bench.py --steps 30 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/prof -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 1 --no-cpu --no-brackets --batch-streams 0 > $OUT/prof_bench.json 2> $OUT/prof.err
export G16_SERIAL_MSM=1
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $OUT/pmc1 -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-brackets --batch-streams 0 > /dev/null 2> $OUT/pmc1.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum -d $OUT/pmc2 -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-brackets --batch-streams 0 > /dev/null 2> $OUT/pmc2.err
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES -d $OUT/pmc3 -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu --no-brackets --batch-streams 0 > /dev/null 2> $OUT/pmc3.err
unset G16_SERIAL_MSM
cd $GRAFT_REPO_ROOT
# device timelines (HIP events, G16_TRACE_HOST): the product schedule and everything-on-one-stream (standalone stages)
G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_conc.err
G16_SERIAL_MSM=1 G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_serial.err
( echo "# device timeline, default bench workload, averaged over 5 proofs (tools/trace_phases.py on G16_TRACE_HOST=1 output): phase durations in ms"; echo "# concurrent (product schedule):"; python tools/trace_phases.py $OUT/trace_conc.err 3; echo "# everything on one stream (G16_SERIAL_MSM=1): standalone stage durations (the G2 lane's dup-row stage still overlaps its reduce)"; python tools/trace_phases.py $OUT/trace_serial.err 3 ) > $OUT/device_timeline.txt
# the example circuit, for continuity (the 1.7 M upper bracket is a leg of the default line: upper_bracket)
timeout -k 10 400 python bench.py --circuit nzcp_example --steps 30 --warmup 5 --batch-proofs 256 --no-cpu --no-plonk --no-brackets > $OUT/bench_example.json 2> $OUT/bench_example.err
find $OUT -name "*.csv" | head -20
tail -c 600 $OUT/bench.json

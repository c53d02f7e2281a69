#!/bin/bash
# round-3 first GPU pass: full -m gpu suite, then the device timelines of the default workload
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03a
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log
grep -q "rc=0" $OUT/pytest.log || exit 1
timeout -k 10 300 python bench.py --steps 20 --warmup 3 --no-cpu --no-plonk --batch-proofs 256 > $OUT/bench.json 2> $OUT/bench.err || exit 1
G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu --batch-streams 0 > /dev/null 2> $OUT/trace_conc.err || exit 1
G16_SERIAL_MSM=1 G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu --batch-streams 0 > /dev/null 2> $OUT/trace_serial.err || exit 1
( echo "# concurrent:"; python tools/trace_phases.py $OUT/trace_conc.err 3; echo "# serial:"; python tools/trace_phases.py $OUT/trace_serial.err 3 ) > $OUT/device_timeline.txt
cat $OUT/device_timeline.txt
python -c "
import json;d=json.load(open('$OUT/bench.json'));print({k:d[k] for k in ('value','ms_per_step','phases_ms')}, d.get('batch_throughput'))"

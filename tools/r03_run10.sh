#!/bin/bash
cd $GRAFT_REPO_ROOT
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03j
rm -rf $OUT && mkdir -p $OUT
timeout -k 10 600 python -m pytest tests/test_gpu_f29.py tests/test_gpu_prove.py tests/test_gpu_edges.py tests/test_gpu_layers.py "tests/test_gpu_fullsize.py::test_real_nzcp_circuit_with_in_circuit_cbor_search" "tests/test_gpu_fullsize.py::test_config5_sha256_chain_real_circuit" -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?"; tail -3 $OUT/pytest.log
grep -q " passed" $OUT/pytest.log || exit 1
bash tools/r03_ab.sh r03j -- base X=1 -- base2 X=2
bash tools/r03_ab.sh r03j --circuit synthetic -- syn_base X=1
timeout -k 10 300 python bench.py --steps 10 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-proofs 512 > $OUT/batch.json 2> $OUT/batch.err
python - <<PY
import json
d=json.load(open("$OUT/batch.json")); print("batch", d["batch_throughput"]["proofs_per_sec"], d["ms_per_step"])
PY
export G16_SERIAL_MSM=1
G16_TRACE_HOST=1 timeout -k 10 200 python bench.py --steps 8 --warmup 3 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_serial.err
echo "# serial"; python tools/trace_phases.py $OUT/trace_serial.err 3
G16_TRACE_HOST=1 timeout -k 10 300 python bench.py --sha256-blocks 163 --steps 6 --warmup 2 --no-cpu --no-plonk --no-brackets --batch-streams 0 > /dev/null 2> $OUT/trace_serial_2p22.err
echo "# serial sha256x163 (2^22)"; python tools/trace_phases.py $OUT/trace_serial_2p22.err 2

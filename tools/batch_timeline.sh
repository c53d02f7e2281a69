# kernel trace of g16_prove_batch (run through gpurun from the repo root): the product's queue map, and every context's
# busy streams at high priority as in the first version (G16_CTX_SAME_PRIO=1)
OUT=$GRAFT_REPO_ROOT/gpurun_out/btl; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/pools -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu --no-plonk --batch-proofs 32 > $OUT/pools.json 2> $OUT/pools.err
export G16_CTX_SAME_PRIO=1
timeout -k 10 300 rocprofv3 --kernel-trace -d $OUT/same -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 2 --no-cpu --no-plonk --batch-proofs 32 > $OUT/same.json 2> $OUT/same.err
unset G16_CTX_SAME_PRIO
cd $GRAFT_REPO_ROOT
( echo "# g16_prove_batch of 32 proofs under rocprofv3 --kernel-trace (tools/batch_timeline.sh + batch_timeline.py)"; echo "## three contexts on their own hardware-queue pools (product)"; python tools/batch_timeline.py $OUT/pools/run_kernel_trace.csv; echo "## three contexts, every busy stream at high priority (G16_CTX_SAME_PRIO=1: the first version's queue map)"; python tools/batch_timeline.py $OUT/same/run_kernel_trace.csv ) > $OUT/summary.txt
cat $OUT/summary.txt

cd /tmp && export TMPDIR=/tmp
timeout -k 10 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/plonk_prof -o run --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/plonk_bench.py --steps 3 --warmup 1 --no-verify > $GRAFT_REPO_ROOT/gpurun_out/plonk_prof.json 2> $GRAFT_REPO_ROOT/gpurun_out/plonk_prof.err

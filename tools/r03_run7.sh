#!/bin/bash
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03g
rm -rf $OUT && mkdir -p $OUT
cd $GRAFT_REPO_ROOT
bash tools/r03_ab.sh r03g -- base X=1 -- occ5 G16_ACC_OCC=5 -- base2 X=2 -- occ5b G16_ACC_OCC=5
bash tools/r03_ab.sh r03g --circuit synthetic -- syn_base X=1 -- syn_occ5 G16_ACC_OCC=5
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1
echo "pytest rc=$?" | tee -a $OUT/pytest.log
tail -5 $OUT/pytest.log

#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/r03_ab.sh r03f -- base X=1 -- prio1 G16_CHAIN_PRIO=1 -- prio2 G16_CHAIN_PRIO=2 -- prio3 G16_CHAIN_PRIO=3 -- hseg4 G16_SEG_LEN=8,4,4 -- accw220 G16_ACC_WAVES=220 -- accw110 G16_ACC_WAVES=110 -- base2 X=2
bash tools/r03_ab.sh r03f --circuit synthetic -- syn_base X=1 -- syn_prio1 G16_CHAIN_PRIO=1 -- syn_prio2 G16_CHAIN_PRIO=2 -- syn_prio3 G16_CHAIN_PRIO=3 -- syn_accw220 G16_ACC_WAVES=220 -- syn_accw110 G16_ACC_WAVES=110 -- syn_p2w2 G16_CHAIN_PRIO=2 G16_ACC_WAVES=220

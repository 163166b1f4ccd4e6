#!/bin/bash
# PMC of one fine-tuning step (three separate passes; kernel-trace only beside the counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r04j; mkdir -p $O
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout 400 rocprofv3 --kernel-trace --pmc $grp -d /tmp/tpmc$i -o p$i -- python bench.py --train --steps 1 --warmup 1 --no-autotune > $O/pmc$i.log 2>&1
done
python tools/pmc_summary.py $O/r04j_train_step_pmc.csv $(find /tmp/tpmc1 /tmp/tpmc2 /tmp/tpmc3 -name "*results.db")
head -14 $O/r04j_train_step_pmc.csv | cut -c1-260

#!/bin/bash
# run on the GPU box: steady-state kernel split of the reverse pass (round 6).  rocprofv3 --kernel-trace --stats of 1 and of 4
# evaluations of the attend-and-excite gradient (B = 1) / the training step's loss + gradients (B = 4) at 512 x 512, eager launches;
# tools/prof_diff.py subtracts them (the first evaluation also packs the backward weight layouts).  -> gpurun_out/prof_{aae,train}_ss.txt
cd /tmp && export TMPDIR=/tmp
for m in aae train; do for r in 1 4; do
  rm -rf /tmp/prof_${m}_$r
  rocprofv3 --kernel-trace --stats -d /tmp/prof_${m}_$r -o p -- python $GRAFT_REPO_ROOT/tools/prof_aae.py $m $r > /tmp/prof_$m.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/prof_${m}_$r -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/prof_${m}_${r}.csv
done; done
python $GRAFT_REPO_ROOT/tools/prof_diff.py $GRAFT_REPO_ROOT/gpurun_out/prof_aae_1.csv 1 $GRAFT_REPO_ROOT/gpurun_out/prof_aae_4.csv 4 > $GRAFT_REPO_ROOT/gpurun_out/prof_aae_ss.txt
python $GRAFT_REPO_ROOT/tools/prof_diff.py $GRAFT_REPO_ROOT/gpurun_out/prof_train_1.csv 1 $GRAFT_REPO_ROOT/gpurun_out/prof_train_4.csv 4 > $GRAFT_REPO_ROOT/gpurun_out/prof_train_ss.txt

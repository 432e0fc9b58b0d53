cd /tmp && export TMPDIR=/tmp
for m in aae train; do for r in 1 4; do
  rm -rf /tmp/prof_${m}_$r
  rocprofv3 --kernel-trace --stats -d /tmp/prof_${m}_$r -o p -- python $GRAFT_REPO_ROOT/tools/prof_aae.py $m $r > /tmp/prof_$m.log 2>&1
  python $GRAFT_REPO_ROOT/tools/rocpd_summary.py $(find /tmp/prof_${m}_$r -name "*.db" | head -1) > $GRAFT_REPO_ROOT/gpurun_out/prof_${m}_${r}.csv
done; done

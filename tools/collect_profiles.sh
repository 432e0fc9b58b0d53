#!/bin/bash
# run on the GPU box: bench line + rocprofv3 kernel stats (timed regime and single-stream regime) + PMC traffic
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
O=$R/gpurun_out/r01; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
# 1. the bench line (default flags)
(cd $R && python bench.py 2>$O/bench.err | tail -1 > $O/bench.json)
# 2. kernel trace + stats of the bench command (graphs, 2 batches in flight)
rm -rf /tmp/rpA; rocprofv3 --kernel-trace --stats -d /tmp/rpA -o t -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_under_rocprof_inflight.json
python $R/tools/rocpd_summary.py $(find /tmp/rpA -name "*.db" | head -1) > $O/kernel_stats_inflight.csv
# 3. the regime the roofline events are taken in: eager launches, one stream, one batch at a time
rm -rf /tmp/rpB; UDT_GRAPHS=0 UDT_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/rpB -o t -- python $R/bench.py --steps 1 --warmup 1 --in-flight 1 --no-cpu-baseline 2>/dev/null | tail -1 > $O/bench_under_rocprof_single.json
python $R/tools/rocpd_summary.py $(find /tmp/rpB -name "*.db" | head -1) > $O/kernel_stats_single.csv
# 4. HBM traffic of the 3x3-conv kernels: two PMC passes (no tracing domains) over ONE batch of the bench workload
#    (tools/predict_once.py: the same launch population as bench.py's roofline pass; bench.py itself crashes
#    rocprofv3's counter collection once its per-launch HIP events are enabled)
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  UDT_GRAPHS=0 UDT_DUAL_STREAM=0 timeout 900 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_$c -o p -- python $R/tools/predict_once.py 50 > /dev/null 2>&1
done
python $R/tools/pmc_traffic.py $(find /tmp/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1) $(find /tmp/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1) > $O/traffic.json
ls -la $O; cat $O/bench.json | cut -c1-600; cat $O/traffic.json

#!/bin/bash
# run on the GPU box: bench line + rocprofv3 kernel stats (timed regime and single-stream regime) + PMC traffic + MFMA util + the
# per-round yardsticks.   usage: tools/collect_profiles.sh <round-tag> <git HEAD sha>   -> gpurun_out/<tag>/
# Every file it writes names the commit it was measured at (first line "# HEAD <sha>" for text / csv, a "head" field for json):
# the box has no .git, so the caller passes `git rev-parse HEAD` (copy what should be judged into profiles/ afterwards).
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=/root/repo
TAG=${1:-r04}
export UDT_HEAD=${2:-unknown}
O=$R/gpurun_out/$TAG; mkdir -p $O
stamp() { for f in "$@"; do [ -f "$f" ] && sed -i "1i # HEAD $UDT_HEAD" "$f"; done; }
jstamp() { for f in "$@"; do [ -s "$f" ] && python - "$f" <<PY
import json, sys
p = sys.argv[1]
d = json.load(open(p)); d["head"] = "$UDT_HEAD"; json.dump(d, open(p, "w"))
PY
done; }
# 0. the GPU test suite and the smoke test at this commit
(cd $R && python -m pytest tests -m gpu -q 2>&1 | grep -E "passed|failed|error" | tail -3 > $O/gpu_tests.log; python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep smoke >> $O/gpu_tests.log)
stamp $O/gpu_tests.log
cd /tmp && export TMPDIR=/tmp
# 1. the bench line (default flags = what the driver runs)
(cd $R && python bench.py 2>$O/bench.err | tail -1 > $O/bench.json); jstamp $O/bench.json
# 2. kernel trace + stats of the bench command (hipGraph replay, three batches in flight)
rm -rf /tmp/rpA; rocprofv3 --kernel-trace --stats -d /tmp/rpA -o t -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-extra-configs --no-mode-table --no-reference-default 2>/dev/null | tail -1 > $O/bench_under_rocprof_inflight.json
python $R/tools/rocpd_summary.py $(find /tmp/rpA -name "*.db" | head -1) > $O/kernel_stats_inflight.csv
# 3. the regime the roofline events are taken in: eager launches, one stream, one batch at a time
rm -rf /tmp/rpB; UDT_GRAPHS=0 UDT_DUAL_STREAM=0 rocprofv3 --kernel-trace --stats -d /tmp/rpB -o t -- python $R/bench.py --steps 1 --warmup 1 --in-flight 1 --fuse 1 --no-cpu-baseline --no-extra-configs --no-mode-table --no-reference-default 2>/dev/null | tail -1 > $O/bench_under_rocprof_single.json
python $R/tools/rocpd_summary.py $(find /tmp/rpB -name "*.db" | head -1) > $O/kernel_stats_single_stream.csv
jstamp $O/bench_under_rocprof_inflight.json $O/bench_under_rocprof_single.json; stamp $O/kernel_stats_inflight.csv $O/kernel_stats_single_stream.csv
# 4. HBM traffic of the 3x3-conv kernels: PMC passes (no tracing domains) over one batch of the bench workload.
#    rocprofv3's counter collection dies after ~6000 dispatches on this image, so a 2-step and a 10-step batch are
#    profiled and extrapolated to 50 steps (tools/pmc_extrapolate.py)
for n in 2 10; do for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_${c}_$n
  UDT_GRAPHS=0 UDT_DUAL_STREAM=0 timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmc_${c}_$n -o p -- python $R/tools/predict_once.py $n > /dev/null 2>&1
done; done
python $R/tools/pmc_extrapolate.py /tmp/pmc_FETCH_SIZE_2/p_counter_collection.csv /tmp/pmc_WRITE_SIZE_2/p_counter_collection.csv /tmp/pmc_FETCH_SIZE_10/p_counter_collection.csv /tmp/pmc_WRITE_SIZE_10/p_counter_collection.csv > $O/traffic.json
#    ... and the same with every launch PLANNED for three batches in flight (the headline mode's split-K / tile plans: round 6's
#    share-aware split-K removes slices there)
for n in 2 10; do for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmci_${c}_$n
  UDT_PLAN_SHARE=3 UDT_GRAPHS=0 UDT_DUAL_STREAM=0 timeout 600 rocprofv3 --pmc $c --output-format csv -d /tmp/pmci_${c}_$n -o p -- python $R/tools/predict_once.py $n > /dev/null 2>&1
done; done
python $R/tools/pmc_extrapolate.py /tmp/pmci_FETCH_SIZE_2/p_counter_collection.csv /tmp/pmci_WRITE_SIZE_2/p_counter_collection.csv /tmp/pmci_FETCH_SIZE_10/p_counter_collection.csv /tmp/pmci_WRITE_SIZE_10/p_counter_collection.csv > $O/traffic_inflight_plans.json
# 5. MFMA utilisation per kernel (one PMC pass over a 4-step batch)
rm -rf /tmp/pmc_mfma
UDT_GRAPHS=0 UDT_DUAL_STREAM=0 timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -o p -- python $R/tools/predict_once.py 4 > /dev/null 2>&1
python $R/tools/pmc_mfma.py /tmp/pmc_mfma/p_counter_collection.csv > $O/mfma_util.json
# 6. config #4 (768x768, batch 8, 12 characters) and the fp8-linears mode (config #5's arithmetic on one GPU)
(cd $R && python bench.py --size 768 --batch 8 --chars 12 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-reference-default 2>/dev/null | tail -1 > $O/bench_config4_768.json)
(cd $R && python bench.py --fp8 --no-cpu-baseline --no-extra-configs --no-reference-default 2>/dev/null | tail -1 > $O/bench_fp8.json)
jstamp $O/bench_config4_768.json $O/bench_fp8.json
# 7. launch-mode sweep on this box: batches in flight
(cd $R && for n in 1 2 3 4 5 6; do python bench.py --in-flight $n --steps 12 --warmup 3 --no-cpu-baseline --no-extra-configs --no-mode-table --no-reference-default 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in_flight $n: %.3f images/s, %.1f ms per batch of 4' % (d['value'], d['ms_per_step']))"; done > $O/in_flight_sweep.txt)
(cd $R && python tools/phase_times.py 2>/dev/null | grep -E "alone|predict_many|sampling only" > $O/phase_times.txt)
# 8. per-shape tables: GEMM shapes (lean family vs the 8-wave kernels), wide vs lean convolution, trace of one step, op micro-benchmarks
(cd $R && python tools/bench_gemm_shapes.py lean=0 lean=-1 2>/dev/null > $O/gemm_shapes.txt)
(cd $R && python tools/bench_wide_conv.py 2>/dev/null > $O/wide_conv.txt)
(cd $R && UDT_DUAL_STREAM=0 python tools/trace_step.py 2>/dev/null > $O/trace_step.txt)
(cd $R && python tools/bench_ops.py 2>/dev/null > $O/bench_ops.txt)
(cd $R && python tools/bench_attn512.py 2>/dev/null > $O/attn512.txt)
(cd $R && python tools/bench_rowres.py 2>/dev/null | grep " x \|shape" > $O/rowres_bench.txt)
(cd $R && python tools/bench_tattn.py 2>/dev/null | grep tattn > $O/tattn_bench.txt)
(cd $R && python tools/bench_reference_default.py 2>/dev/null | tail -1 > $O/reference_default.txt)
# 9. config #5 (round 5): the MX8 linears per layer against the bf16 launches they replace, one eager step traced in fp8 mode, and the
#    MFMA utilisation of the fp8 step
(cd $R && python tools/bench_mx8.py 2>/dev/null > $O/mx8_layers.txt)
(cd $R && UDT_FP8=1 UDT_DUAL_STREAM=0 python tools/trace_step.py 2>/dev/null > $O/trace_step_fp8.txt)
#    ... the e4m3 self-attention next to the bf16 flash kernel (q|k|v projection + attention per level; time against the workgroup count)
#    and the bench line of config #5 with the bf16 attention kept (UDT_FP8_ATTN=0: the MX8 linears alone)
(cd $R && python tools/bench_attn8.py 2>/dev/null | grep -v amdgpu.ids > $O/attn_mx8_layers.txt)
(cd $R && python tools/attn_tail.py 2>/dev/null | grep "workgroups" > $O/attn_vs_workgroups.txt)
(cd $R && UDT_FP8_ATTN=0 python bench.py --fp8 --no-cpu-baseline --no-extra-configs --no-reference-default 2>/dev/null | tail -1 > $O/bench_fp8_linears_only.json)
jstamp $O/bench_fp8_linears_only.json
#    round 6: attend-and-excite gradient and one training step at the benchmark's latent size
(cd $R && python tools/bench_aae.py 2>/dev/null | grep -v amdgpu.ids > $O/aae_training_bench.txt)
stamp $O/aae_training_bench.txt
#    ... and the steady-state kernel split of the two (rocprofv3 --kernel-trace --stats, 4 evaluations minus 1)
bash $R/tools/prof_reverse_pass.sh > /dev/null 2>&1
(echo "## attend-and-excite gradient, 512x512, B = 1 (eager launches under the profiler)"; head -40 $R/gpurun_out/prof_aae_ss.txt; echo; echo "## training step loss + gradients, 512x512, B = 4"; head -48 $R/gpurun_out/prof_train_ss.txt) > $O/reverse_pass_kernel_split.txt
stamp $O/reverse_pass_kernel_split.txt
stamp $O/mx8_layers.txt $O/trace_step_fp8.txt $O/attn_mx8_layers.txt $O/attn_vs_workgroups.txt
stamp $O/in_flight_sweep.txt $O/phase_times.txt $O/gemm_shapes.txt $O/wide_conv.txt $O/trace_step.txt $O/bench_ops.txt $O/attn512.txt $O/rowres_bench.txt $O/tattn_bench.txt $O/reference_default.txt
ls -la $O; cut -c1-400 $O/bench.json; cat $O/traffic.json | head -40; head -30 $O/mfma_util.json

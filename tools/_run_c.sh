mkdir -p gpurun_out/r03f
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03f/gpu_tests.log 2>&1; tail -5 gpurun_out/r03f/gpu_tests.log
cp gpurun_out/parity_report.txt gpurun_out/r03f/ 2>/dev/null
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03f/bench.json 2> gpurun_out/r03f/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03f/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'ms',d['ms_per_step'],'unet_ms',d['unet_ms_per_sampler_step']); print(d['images_per_s_by_launch_mode'])
print('conv',d['roofline']['achieved'],d['roofline']['avg_launch_us']); print({k:(v['achieved'],v['avg_launch_us']) for k,v in d['roofline_classes'].items()})
PY
UDT_DUAL_STREAM=0 timeout 300 python tools/trace_step.py > gpurun_out/r03f/trace.txt 2>&1; head -45 gpurun_out/r03f/trace.txt | tail -42

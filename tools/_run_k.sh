mkdir -p gpurun_out/r03k
timeout 1800 python -m pytest tests -m gpu -x -q > gpurun_out/r03k/gpu_tests.log 2>&1; tail -6 gpurun_out/r03k/gpu_tests.log
cp gpurun_out/parity_report.txt gpurun_out/r03k/ 2>/dev/null
timeout 600 python tools/check_lean_conv.py 2>&1 | tail -17 > gpurun_out/r03k/conv.txt; cat gpurun_out/r03k/conv.txt
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/r03k/bench.json 2> gpurun_out/r03k/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03k/bench.json').read().strip().splitlines()[-1])
print('value',d['value'],'one',d['value_one_batch'],'unet_ms',d['unet_ms_per_sampler_step']); print(d['images_per_s_by_launch_mode']); print(d['images_per_s_reference_default'])
print('conv',d['roofline']['achieved'],d['roofline']['avg_launch_us']); print({k:(v['achieved'],v['avg_launch_us']) for k,v in d['roofline_classes'].items()})
PY
tail -3 gpurun_out/r03k/bench.err

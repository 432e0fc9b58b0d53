mkdir -p gpurun_out/r03h
for n in 1 2 3 4 5 6; do python bench.py --in-flight $n --steps 12 --warmup 3 --no-cpu-baseline --no-mode-table 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('in_flight $n: %.3f images/s, %.1f ms per batch of 4' % (d['value'], d['ms_per_step']))"; done > gpurun_out/r03h/in_flight_sweep.txt
cat gpurun_out/r03h/in_flight_sweep.txt

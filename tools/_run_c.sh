mkdir -p gpurun_out/r03c
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/r03c/gpu_tests.log 2>&1; tail -8 gpurun_out/r03c/gpu_tests.log
cp gpurun_out/parity_report.txt gpurun_out/r03c/ 2>/dev/null
timeout 600 python bench.py --no-cpu-baseline > gpurun_out/r03c/bench.json 2> gpurun_out/r03c/bench.err; cut -c1-1500 gpurun_out/r03c/bench.json
timeout 600 python tools/bench_gemm_shapes.py lean=0 lean=-1 > gpurun_out/r03c/gemm_shapes.txt 2>&1; tail -24 gpurun_out/r03c/gemm_shapes.txt

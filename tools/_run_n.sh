mkdir -p gpurun_out/r03n
timeout 300 python tools/check_lean.py 6 2>&1 | tail -24
timeout 600 python tools/bench_gemm_shapes.py lean=1 lean=6 lean=1 lean=6 > gpurun_out/r03n/gemm_shapes_256.txt 2>&1; tail -26 gpurun_out/r03n/gemm_shapes_256.txt

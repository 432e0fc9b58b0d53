mkdir -p gpurun_out/r03b
timeout 300 python tools/check_lean.py 1 > gpurun_out/r03b/check_lean.txt 2>&1; tail -18 gpurun_out/r03b/check_lean.txt
timeout 900 python tools/bench_gemm_shapes.py lean=0 lean=1,lean_pf=0 lean=1 lean=1,lean_splitk=1 lean=1,lean_splitk=2 lean=1,lean_splitk=3 > gpurun_out/r03b/gemm_shapes_lean.txt 2>&1; tail -30 gpurun_out/r03b/gemm_shapes_lean.txt

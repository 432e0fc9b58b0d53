mkdir -p gpurun_out/r03j
timeout 300 python tools/check_lean.py 1 > gpurun_out/r03j/check_lean.txt 2>&1; tail -7 gpurun_out/r03j/check_lean.txt
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "full_size_properties or benchmarked_shape" 2>&1 | tail -3
UDT_DUAL_STREAM=0 timeout 300 python tools/trace_step.py > gpurun_out/r03j/trace_pf1.txt 2>&1
UDT_DUAL_STREAM=0 UDT_LEAN_PF=0 TRACE_OUT=trace_step0.csv timeout 300 python tools/trace_step.py > gpurun_out/r03j/trace_pf0.txt 2>&1
grep -h "total traced" gpurun_out/r03j/trace_pf1.txt gpurun_out/r03j/trace_pf0.txt
grep "lean1" gpurun_out/r03j/trace_pf1.txt | head -14; echo; grep "lean1" gpurun_out/r03j/trace_pf0.txt | head -14

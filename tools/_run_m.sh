mkdir -p gpurun_out/r03m
timeout 300 python tools/check_lean.py 1 2>&1 | tail -14
timeout 600 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "tattn or conv or vae or attn" 2>&1 | tail -3
UDT_DUAL_STREAM=0 timeout 300 python tools/trace_step.py > gpurun_out/r03m/trace.txt 2>&1
UDT_DUAL_STREAM=0 UDT_TATTN_SPLIT_WGS=128 TRACE_OUT=t0.csv timeout 300 python tools/trace_step.py > gpurun_out/r03m/trace_split128.txt 2>&1
grep -h "total traced" gpurun_out/r03m/trace.txt gpurun_out/r03m/trace_split128.txt
grep -h "tattn" gpurun_out/r03m/trace.txt; echo; grep -h "tattn" gpurun_out/r03m/trace_split128.txt; echo; grep -h "gemm8\|conv=1" gpurun_out/r03m/trace.txt | head; grep "attn B" gpurun_out/r03m/trace.txt

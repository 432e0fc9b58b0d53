mkdir -p gpurun_out/r03e
timeout 600 python tools/check_lean_conv.py > gpurun_out/r03e/check_lean_conv.txt 2>&1; cat gpurun_out/r03e/check_lean_conv.txt | tail -24
timeout 300 python tools/check_lean.py 1 > gpurun_out/r03e/check_lean.txt 2>&1; tail -8 gpurun_out/r03e/check_lean.txt

mkdir -p gpurun_out/r03e
timeout 900 python tools/check_lean_conv.py > gpurun_out/r03e/check_lean_conv2.txt 2>&1; cat gpurun_out/r03e/check_lean_conv2.txt | tail -36

mkdir -p gpurun_out/r03l
timeout 900 python -m pytest tests/test_ops_gpu.py -m gpu -x -q -k "attention" 2>&1 | tail -5
timeout 300 python tools/bench_ops.py 2>&1 | grep attn
UDT_ATTN_V2=0 timeout 300 python tools/bench_ops.py 2>&1 | grep "row-major"
timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "in_flight_with_noise or benchmarked_shape_vs_oracle or checkpoint_load" 2>&1 | tail -4

for cfg in "" "UDT_LEAN_SPLITK=1"; do
  echo "== $cfg"; env $cfg timeout 600 python -m pytest tests/test_engine_gpu.py -m gpu -x -q -k "full_size_properties or 768_path" 2>&1 | tail -3
done
timeout 900 python tools/check_lean_conv.py 2>&1 | tail -17

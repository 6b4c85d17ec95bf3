timeout 400 python -u -m pytest tests/test_window_gpu.py tests/test_stage_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 500 python -u -m pytest tests/test_n5000_gpu.py -x -q -m gpu -k "bounded or fused" 2>&1 | tail -3

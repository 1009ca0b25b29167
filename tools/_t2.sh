timeout 1200 python -m pytest tests/test_gemm16_gpu.py tests/test_parity_gpu.py tests/test_edge_gpu.py tests/test_bf16_io_gpu.py -m gpu -x -q 2>&1 | tail -3
bash tools/_t.sh

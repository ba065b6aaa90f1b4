python -m pytest tests/test_gpu_dense_row.py tests/test_gpu_coop.py tests/test_gpu_memo.py tests/test_gpu_stepping.py -x -q 2>&1 | tail -15
python tools/throughput_map.py 2>&1 | grep -v amdgpu.ids

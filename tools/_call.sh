python -m pytest tests/test_gpu_robust_dense.py tests/test_gpu_robust.py tests/test_gpu_dense_row.py -x -q 2>&1 | tail -15
python tools/_robust_probe.py 2>&1 | grep -v amdgpu.ids
python tools/_robust_probe.py 40000 12 500 2>&1 | grep -v amdgpu.ids

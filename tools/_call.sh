python -m pytest tests/test_gpu_jit.py tests/test_gpu_row_models.py tests/test_gpu_stepping.py tests/test_gpu_autodiff.py -x -q 2>&1 | tail -15
python tools/jit_c5.py 2>&1 | grep -v amdgpu.ids | tail -6

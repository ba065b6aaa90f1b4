set -x
python -m pytest tests/test_gpu_jit.py tests/test_gpu_row_models.py tests/test_gpu_coop.py -x -q 2>&1 | tail -3
python tools/lf_balance.py 2>&1 | grep -v amdgpu.ids
python tools/probe.py shape 6 1000 f64 71428 2>&1 | grep -v amdgpu.ids
python tools/probe.py shape 6 1000 f32 142857 2>&1 | grep -v amdgpu.ids
python tools/probe.py shape 12 500 f64 76923 2>&1 | grep -v amdgpu.ids

set -x
python -m pytest tests/test_gpu_row_models.py -x -q -k "manifold" 2>&1 | tail -30
python -m pytest tests/test_gpu_jit.py -x -q 2>&1 | tail -3

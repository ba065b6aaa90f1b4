cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r2c
( time python -m pytest tests/test_cpp_adaptor.py -m gpu -x -q ) > gpurun_out/r2c/cpp.txt 2>&1
python -m pytest tests -m gpu -x -q 2>&1 | tail -25 > gpurun_out/r2c/pytest_gpu.txt
tail -8 gpurun_out/r2c/cpp.txt; tail -25 gpurun_out/r2c/pytest_gpu.txt

#!/bin/bash
# Build an experimental variant of the f32 / 3-block kernels (the bench shape) with extra -D flags into
# tinyopt_amd/_variants/lib_<tag>.so, reusing the other objects of the normal build.
# usage: tools/variant_build.sh <tag> [-DFOO ...]     then run with TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_<tag>.so
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p tinyopt_amd/_variants
obj=tinyopt_amd/_variants/inst_0_3_$tag.o
sobj=tinyopt_amd/_variants/solve_0_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DTOA_INST_DT=0 -DTOA_INST_NBM=3 "$@" \
  -c tinyopt_amd/csrc/inst.hip -o $obj &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DTOA_INST_DT=0 -DTOA_INST_SOLVE -DTOA_INST_NBM=0 "$@" \
  -c tinyopt_amd/csrc/inst.hip -o $sobj &
wait
others=$(ls tinyopt_amd/csrc/_obj/*.o | grep -v "inst_0_3.o\|solve_0.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $obj $sobj -o tinyopt_amd/_variants/lib_$tag.so
python tools/isa_lint.py $obj | tail -1
python tools/kernel_regs.py $obj "lm_fused_kernelINS_13DenseRowModelIfLi3ELi3|accumulate_kernelINS_13DenseRowModelIfLi3ELi3"
echo built tinyopt_amd/_variants/lib_$tag.so

#!/bin/bash
# Build an experimental variant of one (dtype, block count) translation unit — default f32 / 3 blocks, the bench shape;
# TOA_VARIANT_DT=1 TOA_VARIANT_NBM=1 for the C3 shape — with extra -D flags into
# tinyopt_amd/_variants/lib_<tag>.so, reusing the other objects of the normal build.
# usage: tools/variant_build.sh <tag> [-DFOO ...]     then run with TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_<tag>.so
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
DT=${TOA_VARIANT_DT:-0}; NBM=${TOA_VARIANT_NBM:-3}
mkdir -p tinyopt_amd/_variants
obj=tinyopt_amd/_variants/inst_${DT}_${NBM}_$tag.o
sobj=tinyopt_amd/_variants/solve_${DT}_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DTOA_INST_DT=$DT -DTOA_INST_NBM=$NBM "$@" \
  -c tinyopt_amd/csrc/inst.hip -o $obj &
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -DTOA_INST_DT=$DT -DTOA_INST_SOLVE -DTOA_INST_NBM=0 "$@" \
  -c tinyopt_amd/csrc/inst.hip -o $sobj &
wait
others=$(ls tinyopt_amd/csrc/_obj/*.o | grep -v "inst_${DT}_${NBM}.o\|solve_${DT}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl $others $obj $sobj -o tinyopt_amd/_variants/lib_$tag.so
python tools/isa_lint.py $obj | tail -1
python tools/kernel_regs.py $obj "lm_fused_kernelINS_13DenseRowModelIfLi3ELi3|accumulate_kernelINS_13DenseRowModelIfLi3ELi3"
echo built tinyopt_amd/_variants/lib_$tag.so

#!/bin/bash
# Build an experimental variant of csrc/large_n.hip (the n > 128 pipeline and large_chol_solve_kernel) with extra -D flags into
# tinyopt_amd/_variants/lib_<tag>.so, reusing the other objects of the normal build.
# usage: tools/variant_large.sh <tag> [-DFOO ...]     then run with TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_<tag>.so
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p tinyopt_amd/_variants
obj=tinyopt_amd/_variants/large_n_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c tinyopt_amd/csrc/large_n.hip -o $obj
others=$(ls tinyopt_amd/csrc/_obj/*.o | grep -v "/large_n.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl $others $obj -o tinyopt_amd/_variants/lib_$tag.so
python tools/isa_lint.py $obj | tail -1
python tools/kernel_regs.py $obj "large_chol_solve"
echo built tinyopt_amd/_variants/lib_$tag.so

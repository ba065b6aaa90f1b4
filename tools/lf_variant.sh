#!/bin/bash
# Build an experimental variant of the large-n fused kernel TU (large_fused.hip) with extra -D flags into
# tinyopt_amd/_variants/lib_<tag>.so, reusing the other objects of the normal build.
# usage: tools/lf_variant.sh <tag> [-DFOO ...]     then run with TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_<tag>.so
set -e
cd "$(dirname "$0")/.."
tag=$1; shift
mkdir -p tinyopt_amd/_variants
obj=tinyopt_amd/_variants/large_fused_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c tinyopt_amd/csrc/large_fused.hip -o $obj
others=$(ls tinyopt_amd/csrc/_obj/*.o | grep -v "/large_fused.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $others $obj -ldl -o tinyopt_amd/_variants/lib_$tag.so
python tools/isa_lint.py $obj | tail -1
python tools/kernel_regs.py $obj "large_fused_kernel" 2>/dev/null | head -12
echo built tinyopt_amd/_variants/lib_$tag.so

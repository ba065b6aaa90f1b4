#!/bin/bash
# Build an experimental variant of ONE translation unit with extra -D flags into tinyopt_amd/_variants/lib_<tag>.so, reusing the
# other objects of the normal build.   usage: tools/variant_tu.sh <tag> <source in csrc/> <object name in csrc/_obj/> [-DFOO ...]
#   e.g. tools/variant_tu.sh blt ba_schur.hip ba_schur.o -DTOA_BL_TIMING     then TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_blt.so
set -e
cd "$(dirname "$0")/.."
tag=$1; src=$2; objname=$3; shift 3
mkdir -p tinyopt_amd/_variants
obj=tinyopt_amd/_variants/${objname%.o}_$tag.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize "$@" -c tinyopt_amd/csrc/$src -o $obj
others=$(ls tinyopt_amd/csrc/_obj/*.o | grep -v "/$objname")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -ldl $others $obj -o tinyopt_amd/_variants/lib_$tag.so
python tools/isa_lint.py $obj | tail -1
echo built tinyopt_amd/_variants/lib_$tag.so

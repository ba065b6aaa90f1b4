#!/bin/bash
# usage: tools/chol_ab.sh <variant tags...>   (run on the GPU box; -DTOA_CHOL_TIMING builds print their phases)
for v in "$@"; do
  echo "== $v"
  TINYOPT_AMD_LIB=$PWD/tinyopt_amd/_variants/lib_$v.so REPS=${REPS:-1} SHAPES=${SHAPES:-2} python tools/chol_probe.py 2>&1 | grep -av "amdgpu.ids" | tr -s '\n' > /tmp/chol_$v.txt
  grep -a "per call" /tmp/chol_$v.txt
  grep -a "^chol n=384" /tmp/chol_$v.txt | tail -3; grep -a "^chol n=256" /tmp/chol_$v.txt | tail -3
done

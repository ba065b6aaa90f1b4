export ROWBENCH_ONLY="text+J"
for F in "" "-DTOA_ROW_SINGLE=1" "-DTOA_ROW_NBUF=3" "-DTOA_ROW_WAVES=3 -DTOA_ROW_SINGLE=1" "-DTOA_ROW_NBUF=2 -DTOA_ROW_WAVES=3"; do
  echo "FLAGS: $F"; TOA_JIT_FLAGS="$F" python tools/row_model_bench.py 12500 2>&1 | grep "text+J" | sed 's/passes per iteration.*build/build/'
done
export ROWBENCH_ONLY="text AD"
for F in "" "-DTOA_ROW_WAVES=3"; do
  echo "FLAGS: $F"; TOA_JIT_FLAGS="$F" python tools/row_model_bench.py 12500 2>&1 | grep "text AD" | sed 's/passes per iteration.*build/build/'
done

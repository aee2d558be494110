#!/bin/bash
# ab_encode.sh <out-file> <kinds...> : time alpgpu_encode_vectors for every library under build/variants (and the default build)
out=$1; shift
kinds="$@"
: > "$out"
for lib in alp_amd/libalpgpu.so build/variants/libalpgpu_*.so; do
  for k in $kinds; do
    ALPGPU_LIB=$PWD/$lib timeout 300 python tools/time_vectors.py $k 1048576 2>&1 | grep -v "^k histogram" >> "$out"
  done
done
cat "$out"

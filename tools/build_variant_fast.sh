#!/bin/bash
# build_variant_fast.sh <name> <file.hip> [-D...]: an A/B build of libalpgpu.so in which ONE source is compiled with extra flags and linked with the
# objects of the default build (build/obj/, made by __graft_entry__.build()) -> build/variants/libalpgpu_<name>.so (select with ALPGPU_LIB=...)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; file=$2; shift 2
mkdir -p "$ROOT/build/variants"
obj="$ROOT/build/variants/${name}_${file%.hip}.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC "$@" -c "$ROOT/alp_amd/csrc/$file" -o "$obj"
others=$(ls "$ROOT"/build/obj/*.o | grep -v "/${file%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$ROOT/build/variants/libalpgpu_$name.so" "$obj" $others
rm -f "$obj"
python3 "$ROOT/tools/check_top_vgpr.py" --library "$ROOT/build/variants/libalpgpu_$name.so" | tail -1
ls -la "$ROOT/build/variants/libalpgpu_$name.so"

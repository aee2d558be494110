#!/bin/bash
# build_variant.sh <name> [-D...]: an A/B build of libalpgpu.so -> build/variants/libalpgpu_<name>.so (select with ALPGPU_LIB=...)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
mkdir -p "$ROOT/build/variants"
srcs=""
for s in api_context.hip api_encode.hip api_decode.hip api_primitives.hip api_container.hip api_host.hip decode_kernels.hip encode_kernels.hip encode_lean_kernels.hip init_kernels.hip primitive_kernels.hip pad_kernels.hip decode_f32_kernels.hip encode_f32_kernels.hip primitive_f32_kernels.hip consume_kernels.hip guard_kernels.hip read_ahead_kernels.hip; do srcs="$srcs $ROOT/alp_amd/csrc/$s"; done
timeout 1200 /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math -fPIC -shared "$@" -o "$ROOT/build/variants/libalpgpu_$name.so" $srcs 2>&1 | grep -v hip-link || true
ls -la "$ROOT/build/variants/libalpgpu_$name.so"
# every build route checks its artefact for the gfx950 last-register pattern (tools/check_top_vgpr.py --library)
python3 "$ROOT/tools/check_top_vgpr.py" --library "$ROOT/build/variants/libalpgpu_$name.so" | tail -1

#!/bin/bash
# usage: tools/build_variant.sh <name> [extra hipcc flags...]  -> lz4-java_amd/variants/<name>.so (developer A/B builds, git-ignored)
here="$(cd "$(dirname "$0")/.." && pwd)"
name=$1; shift
mkdir -p "$here/lz4-java_amd/variants"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 "$@" -fPIC -shared -fvisibility=hidden -Wl,-rpath,/opt/rocm/lib -Wl,--exclude-libs,ALL \
  "$here/lz4-java_amd/csrc/kernels.hip" "$here/lz4-java_amd/csrc/api.cpp" -o "$here/lz4-java_amd/variants/$name.so" 2>&1 | grep -v "warning\|^$\|note:" | grep -A6 "error" | head -20
echo "built variants/$name.so"

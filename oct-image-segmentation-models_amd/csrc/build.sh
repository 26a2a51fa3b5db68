#!/bin/bash
# Build liboct_unet_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
OUT=../liboct_unet_hip.so
SRC=oct_unet.hip
if [ -f "$OUT" ] && [ -z "$(find . ../../include -newer "$OUT" -type f \( -name '*.hip' -o -name '*.hpp' -o -name '*.h' -o -name 'build.sh' \) | head -1)" ]; then
    exit 0   # up to date
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    "$@" -o "$OUT.tmp" "$SRC"
mv "$OUT.tmp" "$OUT"

#!/bin/bash
# Build liboct_unet_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [extra hipcc flags]
set -euo pipefail
cd "$(dirname "$0")"
OUT=../liboct_unet_hip.so
SRC=oct_unet.hip
if [ -f "$OUT" ] && [ -f ../liboct_minpath.so ] && [ ! minpath.cpp -nt ../liboct_minpath.so ] && [ -z "$(find . ../../include -newer "$OUT" -type f \( -name '*.hip' -o -name '*.hpp' -o -name '*.h' -o -name 'build.sh' \) | head -1)" ]; then
    exit 0   # up to date
fi
# host-side native code: min-path delineation (plain C++, no GPU)
if [ ! -f ../liboct_minpath.so ] || [ minpath.cpp -nt ../liboct_minpath.so ]; then
    g++ -O3 -std=c++17 -fPIC -shared -Wall -o ../liboct_minpath.so.tmp minpath.cpp && mv ../liboct_minpath.so.tmp ../liboct_minpath.so
fi
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wall -Wno-unused-function \
    "$@" -o "$OUT.tmp" "$SRC"
mv "$OUT.tmp" "$OUT"

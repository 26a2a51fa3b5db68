#!/bin/bash
# Build liboct_unet_hip.so for gfx950 (cross-compiles without a GPU).  Usage: build.sh [-f] [extra hipcc flags]
# The library is several translation units (host.hpp) compiled in parallel and linked into one .so; objects are kept
# under build/ and only rebuilt when a source they include is newer.  The SHA-256 of the sources is compiled into
# oct_version(): _hip.py refuses a library whose stamp differs from the sources beside it (stale binary).
set -euo pipefail
cd "$(dirname "$0")"
FORCE=0
if [ "${1:-}" = "-f" ]; then FORCE=1; shift; fi
OUT=../liboct_unet_hip.so
HIPCC=/opt/rocm/bin/hipcc
export LC_ALL=C
SRCS="$(ls *.hip *.hpp ../../include/*.h | sort)"
HASH=$(cat $SRCS | sha256sum | cut -c1-12)
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -Wno-unused-but-set-variable $*"
FLAGHASH=$(echo "$FLAGS" | sha256sum | cut -c1-8)

# host-side native code: min-path delineation (plain C++, no GPU)
if [ ! -f ../liboct_minpath.so ] || [ minpath.cpp -nt ../liboct_minpath.so ]; then
    g++ -O3 -std=c++17 -fPIC -shared -Wall -o ../liboct_minpath.so.tmp minpath.cpp && mv ../liboct_minpath.so.tmp ../liboct_minpath.so
fi

mkdir -p build
if [ -f "$OUT" ] && [ "$FORCE" = 0 ] && [ -f build/stamp ] && [ "$(cat build/stamp)" = "$HASH $FLAGHASH" ]; then
    exit 0   # up to date: same sources, same flags
fi
pids=()
objs=()
for src in *.hip; do
    obj=build/${src%.hip}.o
    objs+=("$obj")
    # (every .hip includes most headers: any newer header rebuilds everything; oct_unet.hip also carries the source stamp)
    if [ "$FORCE" = 1 ] || [ ! -f "$obj" ] || [ ! -f build/flags ] || [ "$(cat build/flags)" != "$FLAGHASH" ] || [ "$src" = oct_unet.hip ] || \
       [ -n "$(find . ../../include -newer "$obj" -type f \( -name '*.hpp' -o -name '*.h' \) | head -1)" ] || [ "$src" -nt "$obj" ]; then
        ( $HIPCC $FLAGS -DOCT_SRC_HASH="\"$HASH\"" -c "$src" -o "$obj.tmp" && mv "$obj.tmp" "$obj" ) &
        pids+=($!)
    fi
done
rc=0
for p in "${pids[@]:-}"; do [ -z "$p" ] || wait "$p" || rc=1; done
[ "$rc" = 0 ] || { echo "build.sh: compilation failed" >&2; exit 1; }
echo "$FLAGHASH" > build/flags
$HIPCC --offload-arch=gfx950 -shared -fPIC -o "$OUT.tmp" "${objs[@]}"
mv "$OUT.tmp" "$OUT"
echo "$HASH $FLAGHASH" > build/stamp

#!/bin/bash
# Builds tests/hostsan/capi_harness against the HOST halves of every translation unit of libworld_mi355 (hipcc
# --cuda-host-only: no device code is compiled, nothing of it could run here) with AddressSanitizer + UBSan, and the
# HIP runtime replaced by tests/hostsan/hip_stub.cpp.  usage: build_capi_harness.sh OUT_DIR
set -e
here="$(cd "$(dirname "$0")" && pwd)"; root="$(cd "$here/../.." && pwd)"; out="$1"; mkdir -p "$out"
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
SAN="-fsanitize=address,undefined -fno-gpu-sanitize -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
src="$root/hts-train-world_amd/csrc"
pids=()
for f in cheaptrick.hip d4c.hip stonemask.hip dio.hip synthesis.hip harvest.hip codec.hip status.hip vibrato.hip context.cpp capi.cpp fileio.cpp; do
  $HIPCC -O1 -g -std=c++17 -fPIC --offload-arch=gfx950 --cuda-host-only $SAN -Wno-comment -I"$root/include" -I"$src" -x hip -c "$src/$f" -o "$out/${f%.*}.o" 2> "$out/${f%.*}.log" &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
# every unit references the symbol of its (absent) device image
nm "$out"/*.o | awk '$1=="U" && $2 ~ /^__hip_fatbin_/ {print $2}' | sort -u | awk '{print "extern \"C\" { char " $1 "[16]; }"}' > "$out/fatbin_syms.cpp"
$HIPCC -O1 -g -std=c++17 --cuda-host-only $SAN -I"$root/include" -x hip -c "$here/hip_stub.cpp" -o "$out/hip_stub.o" 2> "$out/hip_stub.log"
$HIPCC -O1 -g -std=c++17 --cuda-host-only $SAN -I"$root/include" -x hip -c "$here/capi_harness.cpp" -o "$out/capi_harness.o" 2> "$out/capi_harness.log"
$HIPCC -O1 -g -std=c++17 --cuda-host-only $SAN -x hip -c "$out/fatbin_syms.cpp" -o "$out/fatbin_syms.o" 2> "$out/fatbin_syms.log"
# plain clang++ link: no libamdhip64 (the stub IS the runtime)
CLANGXX="$(dirname "$(readlink -f "$HIPCC")")/../lib/llvm/bin/clang++"
[ -x "$CLANGXX" ] || CLANGXX=/opt/rocm/lib/llvm/bin/clang++
"$CLANGXX" $SAN -pthread "$out"/*.o -o "$out/capi_harness" -lm

#!/bin/bash
# Build a variant of libgpe.so (extra -D flags on k_native.hip only) into gpurun_tmp/variants/<name>.so; the other
# objects come from the regular in-tree build.  Runs here (hipcc cross-compiles); the .so travels with gpurun.
# usage: bash scripts/build_variant.sh <name> "<flags>" [file.hip ...]   (default file: k_native.hip)
set -eu
name="$1"; flags="${2:-}"; shift; shift || true
files=("$@"); [ ${#files[@]} -eq 0 ] && files=(k_native.hip)
cd "$(dirname "$0")/.."
python gpu-physics-engine_amd/build.py > /dev/null
B=gpu-physics-engine_amd/csrc/build; V=gpurun_tmp/variants; mkdir -p $V/obj_$name
objs=()
for o in $B/*.o; do
  base=$(basename $o .o); skip=0
  for f in "${files[@]}"; do [ "$base.hip" = "$f" ] && skip=1; done
  [ $skip -eq 0 ] && objs+=($o)
done
for f in "${files[@]}"; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -fPIC -ffp-contract=off -fno-fast-math -Wno-unused-function \
    $flags -c gpu-physics-engine_amd/csrc/$f -o $V/obj_$name/${f%.hip}.o
  objs+=($V/obj_$name/${f%.hip}.o)
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $V/$name.so "${objs[@]}" -ldl
echo "built $V/$name.so [$flags]"

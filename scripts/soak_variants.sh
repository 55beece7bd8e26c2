#!/bin/bash
# same box: the 100M gravity-on scene after S steps (scripts/soak_profile.py) for several builds
# usage: bash scripts/soak_variants.sh S "<flags 1>" "<flags 2>" ...   ("-" = no flags)
set -u
S=$1; shift
for v in "$@"; do
  flags="$v"; [ "$v" = "-" ] && flags=""
  GPE_EXTRA_CXXFLAGS="$flags" python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1 || { echo "build failed: $v"; continue; }
  echo "variant [$v]"
  GPE_NATIVE_STATS=1 timeout -k 10 500 python scripts/soak_profile.py 100000000 $S 2>&1 | grep -v "call [0-9]*[0-9][0-9][0-9]: \|call [1-9][0-9]: " | cut -c1-230 | tail -6
done
python gpu-physics-engine_amd/build.py --force > /dev/null 2>&1

#!/bin/bash
# Same box, same build: time the native step with different gpe_config.flags (2 = sort every step, as rounds 1-2 did).
# usage: bash scripts/ab_flags.sh [-q] <flags> ...
set -u
quick=0; [ "$1" = "-q" ] && { quick=1; shift; }
for f in "$@"; do
  echo "flags [$f]"
  for r in 1 2; do timeout -k 10 120 python scripts/time_step.py 1000000 300 off $f 2>&1 | grep "^n=" | cut -c1-260; done
  timeout -k 10 120 python scripts/time_step.py 16000000 60 on $f 2>&1 | grep "^n=" | cut -c1-260
  [ $quick -eq 0 ] && timeout -k 10 200 python scripts/time_step.py 100000000 40 on $f 2>&1 | grep "^n=" | cut -c1-260
done

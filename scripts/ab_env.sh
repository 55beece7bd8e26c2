#!/bin/bash
# same box, same build: time the native step with different environments.  usage: bash scripts/ab_env.sh "VAR=1" "VAR=0" ...
set -u
for v in "$@"; do
  echo "env [$v]"
  for r in 1 2; do env $v timeout -k 10 120 python scripts/time_step.py 1000000 300 2>&1 | grep "^n=" | cut -c1-230; done
  env $v timeout -k 10 120 python scripts/time_step.py 16000000 40 2>&1 | grep "^n=" | cut -c1-230
  env $v timeout -k 10 200 python scripts/time_step.py 100000000 30 on 2>&1 | grep "^n=" | cut -c1-230
done

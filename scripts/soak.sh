#!/bin/bash
# GPU box: long runs of the native step (sticky device errors surface at the syncs of time_evolution.py)
set -u
timeout -k 10 300 python scripts/time_evolution.py 100000000 5 120 on 2>&1 | grep -v amdgpu | cut -c1-250 || exit 1
timeout -k 10 300 python scripts/time_evolution.py 4000000 8 1500 on 2>&1 | grep -v amdgpu | cut -c1-250 || exit 1

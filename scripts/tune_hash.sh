#!/bin/bash
# hash kernel grid size (tuning only)
set -u
for g in 64 122 244 488; do GPE_HASH_GRID=$g timeout -k 10 120 python scripts/time_step.py 1000000 100 || exit 1; done
for g in 256 512 1024 2048; do GPE_HASH_GRID=$g timeout -k 10 120 python scripts/time_step.py 16000000 50 || exit 1; done
for g in 256 512 1024 2048; do GPE_HASH_GRID=$g timeout -k 10 200 python scripts/time_step.py 100000000 20 on || exit 1; done

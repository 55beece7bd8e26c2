#!/bin/bash
# onesweep tile size x look-back window at small particle counts (tuning only)
set -u
for n in 1000000 4000000 16000000; do
  for items in 4 16; do
    for win in 8 32; do
      GPE_OS_ITEMS=$items GPE_OS_WIN=$win timeout -k 10 120 python scripts/time_step.py $n 100 || exit 1
    done
  done
done

#!/bin/bash
# chunk / grid sweep of the frame kernel (debug overrides read by the library)
set -e
for c in 1 2 4 8; do
  for g in 3 4 6 8; do
  echo "== chunk $c grid/CU $g"; RTX_CHUNK=$c RTX_GRID_PER_CU=$g timeout -k 10 100 python -u tools/perf_probe.py --only "headline,looking away,1 spp,close-up d=1.2" --reps 4
  done
done

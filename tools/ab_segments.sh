#!/bin/bash
# separate processes, alternating, on one box: bash tools/ab_segments.sh
for rep in 1 2 3; do
  python tools/ab_segments.py eager
  PDES_FORK_BATCH=2 python tools/ab_segments.py eager
done

#!/bin/bash
# separate processes, alternating, on one box: bash tools/archive/ab_segments.sh
for rep in 1 2 3; do
  python tools/archive/ab_segments.py eager
  PDES_FORK_BATCH=2 python tools/archive/ab_segments.py eager
done

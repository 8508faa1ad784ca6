#!/bin/bash
# separate processes, alternating, on one box: bash tools/ab_segments.sh
for rep in 1 2; do
  python tools/ab_segments.py eager
  python tools/ab_segments.py forward
  python tools/ab_segments.py segments
  PDES_SEG_SPLITW=1 python tools/ab_segments.py segments
  PDES_SEG_MAX=4 PDES_SEG_SPLITW=1 python tools/ab_segments.py segments
  PDES_SEG_MAX=3 python tools/ab_segments.py segments
done

#!/bin/bash
for i in 1 2 3; do
python3 bench.py --steps 100 --warmup 30 --no-cpu-baseline 2>/dev/null | python3 tools/diag/dropin_leg.py
done

#!/bin/bash
# plan-sweep build only (-DPDES_BAND_PLAN_ENV): the band kernel's rate at one size under forced plans
#   bash tools/sweep_band_plan.sh 65x65 "4,1,5 5,2,2 8,1,3"      (plan = waves,npass,nbands)
for P in $2; do
  printf "%-10s " $P
  PDES_BAND_PLAN=$P python tools/bench_loss_generic.py "row bands $1" 2>&1 | grep -v amdgpu.ids | grep -v batch
done

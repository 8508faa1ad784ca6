for kv in "X=1" "PDES_FORK_SIGNAL=0" "PDES_FIN_ONLOAD=0" "PDES_FIN_ONLOAD=1" "X=2"; do
  printf "%-22s " $kv
  env $kv python bench.py --leg cglow --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])"
done

cd $GRAFT_REPO_ROOT
for i in 1 2 3 4 5 6; do python tools/diag/process_repro.py 2>&1 | grep data; done
for t in 1 4 64; do OMP_NUM_THREADS=$t OPENBLAS_NUM_THREADS=$t python tools/diag/process_repro.py 2>&1 | grep data; done
PDES_HOST_THREADS=0 python tools/diag/process_repro.py 2>&1 | grep data

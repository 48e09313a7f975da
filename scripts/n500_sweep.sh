#!/bin/bash
# N = 500: chunk widths (EQF_SYRK_LA_B) and the workers' share of the panels (EQF_SYRK_LA_FRAC, percent) of the in-launch covariance update against the r5 libraries, same box
mkdir -p gpurun_out/$1
run() { EQVIO_AMD_LIB_DIR=$2 timeout 300 python bench.py --landmarks ${NN:-500} --no-pmc --no-cpu-baseline --no-multi-filter --no-frame-mix --no-binding --no-sizes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(1e3*d['ms_per_step'],2))"; }
( run r5 $PWD/scripts/ab_libs_r5
for F in 30 50 60 70 80; do for B in 4 6 8; do EQF_SYRK_LA_FRAC=$F EQF_SYRK_LA_B=$B run tree_F${F}_B$B ""; done; done
run r5 $PWD/scripts/ab_libs_r5 ) | tee gpurun_out/$1/n500_sweep.txt

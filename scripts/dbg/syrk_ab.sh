#!/bin/bash
# N = 500: variants of the covariance update kernel (libraries built with -DSYRK_UNR / -DSYRK_UNR_TALL; EQF_OPTIONS 23 = EQF_OPT_SYRK_TALL_TILES), same box, alternating
run() { EQVIO_AMD_LIB_DIR=$1 EQF_OPTIONS=$2 timeout 300 python bench.py --landmarks ${NN:-500} --steps 1500 --warmup 200 --no-cpu-baseline --no-multi-filter --no-frame-mix --no-binding --no-pmc --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$3', round(d['value']), round(1e3*d['ms_per_step'],2))"; }
for i in 1 2; do
run "" 23=0 "tree 32x32 unroll 4   "
run "" 23=1 "tree 64x32 unroll 4   "
run $PWD/scripts/ab_libs_u8 23=0 "32x32 unroll 8        "
run $PWD/scripts/ab_libs_u2 23=0 "32x32 unroll 2        "
run $PWD/scripts/ab_libs_t8 23=1 "64x32 unroll 8        "
done

#!/bin/bash
# Same-box A/B of two builds of the libraries: the tree's own (eqvio_amd/lib) against a second set in $1 (default scripts/ab_libs_old, built from another
# commit in a git worktree), alternating, N = 200 unless $2 says otherwise. Box-to-box scatter of the headline is 2-3 %: only same-box pairs mean anything.
OLD=${1:-scripts/ab_libs_old}; N=${2:-200}
run() { EQVIO_AMD_LIB_DIR=$1 timeout 300 python bench.py --landmarks $N --no-cpu-baseline --no-multi-filter --no-frame-mix --no-binding --no-sizes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), round(1e3*d['ms_per_step'],2), d['roofline']['per_kernel_us_per_frame'])"; }
for i in 1 2 3; do run "$PWD/$OLD" old; run "" new; done

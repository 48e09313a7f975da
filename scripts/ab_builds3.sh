#!/bin/bash
# Same-box comparison of several builds of the libraries (directories given as arguments; "" = the tree's own), alternating, three rounds. N from $N (default 200).
N=${N:-200}
run() { EQVIO_AMD_LIB_DIR=$1 timeout 300 python bench.py --landmarks $N --no-pmc --no-cpu-baseline --no-multi-filter --no-frame-mix --no-binding --no-sizes 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), round(1e3*d['ms_per_step'],2), d['roofline']['per_kernel_us_per_frame'])"; }
for i in $(seq 1 ${ROUNDS:-3}); do for d in "$@"; do if [ -z "$d" ] || [ "$d" = tree ]; then run "" tree; else run "$PWD/$d" "$d"; fi; done; done

#!/bin/bash
# headline with option sets (EQF_OPTIONS), same box, alternating
run() { EQF_OPTIONS=$1 timeout 300 python bench.py --steps 4000 --warmup 300 --no-cpu-baseline --no-multi-filter --no-frame-mix --no-binding --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(1e3*d['ms_per_step'],2), d['roofline']['per_kernel_us_per_frame'])"; }
for i in 1 2 3; do for o in "$@"; do run $o; done; done

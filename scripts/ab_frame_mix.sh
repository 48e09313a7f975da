#!/bin/bash
# Same-box comparison of the frame-mix lines (landmark turnover; + shipped outlier thresholds) and the binding for several builds of the libraries, alternating, two rounds.
run() { EQVIO_AMD_LIB_DIR=$1 timeout 600 python bench.py --steps 2000 --warmup 200 --no-pmc --no-cpu-baseline --no-multi-filter ${BIND:---no-binding} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$2', round(d['value']), [round(f['value']) for f in d['frame_mix']], d.get('reference_side_binding',{}).get('member_for_member',{}).get('value'))"; }
for i in 1 2; do for d in "$@"; do if [ -z "$d" ] || [ "$d" = tree ]; then run "" tree; else run "$PWD/$d" "$d"; fi; done; done

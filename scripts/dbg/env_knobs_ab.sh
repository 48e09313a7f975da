#!/bin/bash
# runtime knobs on the frame boundary (launch call + launch -> start), same box, alternating
run() { env "$@" timeout 300 python bench.py --landmarks ${NN:-200} --steps 4000 --warmup 300 --no-cpu-baseline --no-multi-filter --no-frame-mix --no-binding --no-pmc --no-roofline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(1e3*d['ms_per_step'],2))"; }
for i in 1 2 3; do
run A=1
run HIP_FORCE_DEV_KERNARG=1
run HSA_ENABLE_INTERRUPT=0
run HIP_FORCE_DEV_KERNARG=1 HSA_ENABLE_INTERRUPT=0
run HIP_FORCE_DEV_KERNARG=0
done

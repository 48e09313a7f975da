run() { timeout 300 env "$@" python bench.py --no-cpu-baseline --no-multi-filter --no-frame-mix --no-binding --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(1e3*d['ms_per_step'],2))"; }
for i in 1 2; do
echo base; run A=1
echo devkernarg1; run HIP_FORCE_DEV_KERNARG=1
echo devkernarg0; run HIP_FORCE_DEV_KERNARG=0
echo nointerrupt; run HSA_ENABLE_INTERRUPT=0
echo hwq1; run GPU_MAX_HW_QUEUES=1
echo sdma0; run HSA_ENABLE_SDMA=0
done

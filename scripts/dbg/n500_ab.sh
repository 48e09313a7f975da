run() { timeout 300 env "$@" python bench.py --landmarks ${NN:-500} --no-cpu-baseline --no-multi-filter --no-frame-mix --no-binding --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(1e3*d['ms_per_step'],2), d['roofline']['per_kernel_us_per_frame'], d['factorisation'])"; }
for i in 1 2; do
echo -n "split off: "; run EQF_OPTIONS=18=0
echo -n "split on : "; run A=1
done

run() { timeout 300 env "$@" python bench.py --no-cpu-baseline --no-multi-filter --no-frame-mix --no-binding --no-pmc 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(round(d['value']), round(1e3*d['ms_per_step'],2))"; }
for i in 1 2 3; do
echo each; run A=1
echo counted; run EQF_DOOR_COUNTED=1
done

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for opt in "A=1" "EQF_OPTIONS=7=0"; do
rm -rf /tmp/p_kt
env $opt rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- python $R/bench.py --landmarks 500 --steps 30 --warmup 5 --no-cpu-baseline --no-roofline --no-multi-filter --no-frame-mix --no-binding > /tmp/kt.log 2>&1
echo "== $opt"; python $R/scripts/rocpd_stats.py $(find /tmp/p_kt -name "*.db" | head -1) | grep -v mfma_peak | head -8 | python -c "import sys,csv
for r in csv.reader(sys.stdin): print(r[0][:44], r[1], r[3])"
done

# The round's collection in two gpurun calls (the full set of scripts/collect_round.sh takes ~40 GPU-minutes): scripts/collect_round_short.sh <tag> A|B
set -x
T=$1
mkdir -p gpurun_out/${T}
if [ "$2" = "A" ]; then
  (timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -5) > gpurun_out/${T}/${T}_gpu_tests.txt
  bash scripts/collect_profiles.sh ${T} 200 full > gpurun_out/${T}_collect.log 2>&1
  python scripts/lookahead_trace.py 200 > gpurun_out/${T}/${T}_lookahead_trace.txt 2>&1
  python bench.py > gpurun_out/${T}/${T}_full_bench.json 2> gpurun_out/${T}/full_bench.err
  tail -3 gpurun_out/${T}/${T}_gpu_tests.txt; tail -c 300 gpurun_out/${T}/${T}_full_bench.json
else
  bash scripts/collect_profiles.sh ${T}_N500 500 > gpurun_out/${T}_N500_collect.log 2>&1
  python scripts/lookahead_trace.py 500 > gpurun_out/${T}_N500/${T}_N500_lookahead_trace.txt 2>&1
  python scripts/soak_sizes.py 1000 > gpurun_out/${T}/${T}_size_sweep.txt 2>&1
  python tests/run_configs.py > gpurun_out/${T}/${T}_configs.json 2> gpurun_out/${T}/configs.err
  cat gpurun_out/${T}/${T}_size_sweep.txt; tail -c 400 gpurun_out/${T}_N500/${T}_N500_bench.json
fi

set -x
T=$1
bash scripts/collect_profiles.sh ${T} 200 full > gpurun_out/${T}_collect.log 2>&1
python scripts/lookahead_trace.py 200 > gpurun_out/${T}/${T}_lookahead_trace.txt 2>&1
python bench.py > gpurun_out/${T}/${T}_full_bench.json 2> gpurun_out/${T}/full_bench.err
bash scripts/collect_profiles.sh ${T}_N500 500 > gpurun_out/${T}_N500_collect.log 2>&1
python scripts/lookahead_trace.py 500 > gpurun_out/${T}_N500/${T}_N500_lookahead_trace.txt 2>&1
bash scripts/collect_profiles.sh ${T}_N50 50 > gpurun_out/${T}_N50_collect.log 2>&1
python tests/run_configs.py > gpurun_out/${T}/${T}_configs.json 2> gpurun_out/${T}/configs.err
python scripts/soak_sizes.py > gpurun_out/${T}/${T}_size_sweep.txt 2>&1
python bench.py --config batch8 > gpurun_out/${T}/${T}_batch8_1rank.json 2>> gpurun_out/${T}/full_bench.err
EQVIO_BENCH_ONE_DEVICE=1 python bench.py --config batch8 --gpus 8 > gpurun_out/${T}/${T}_batch8_8ranks_one_device.json 2>> gpurun_out/${T}/full_bench.err
EQVIO_BENCH_ONE_DEVICE=1 python bench.py --gpus 8 --steps 3000 --warmup 300 > gpurun_out/${T}/${T}_8ranks_one_device.json 2>> gpurun_out/${T}/full_bench.err
tail -c 200 gpurun_out/${T}/${T}_full_bench.json

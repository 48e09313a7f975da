set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-v9}
OUT=$R/gpurun_out/$V
mkdir -p $OUT
BCMD="python $R/bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-roofline --no-multi-filter"
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $BCMD > /tmp/kt.log 2>&1
DB=$(find /tmp/p_kt -name "*.db" | head -1)
python $R/scripts/rocpd_stats.py $DB > $OUT/r01_${V}_kernel_stats.csv
python $R/scripts/rocpd_timeline.py $DB > $OUT/r01_${V}_timeline.txt
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o f -- $BCMD > /tmp/f.log 2>&1
python $R/scripts/rocpd_pmc.py $(find /tmp/p_f -name "*.db" | head -1) > $OUT/r01_${V}_pmc_fetch_size.csv
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o w -- $BCMD > /tmp/w.log 2>&1
python $R/scripts/rocpd_pmc.py $(find /tmp/p_w -name "*.db" | head -1) > $OUT/r01_${V}_pmc_write_size.csv
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_m -o m -- $BCMD > /tmp/m.log 2>&1
python $R/scripts/rocpd_pmc.py $(find /tmp/p_m -name "*.db" | head -1) > $OUT/r01_${V}_pmc_mfma.csv
cd $R
python scripts/frame_trace.py 200 3000 > $OUT/r01_${V}_frame_trace.txt 2>&1
python scripts/host_share.py 200 6000 > $OUT/r01_${V}_host_share.txt 2>&1
python scripts/propagate_vs_steps.py > $OUT/r01_${V}_propagate_vs_steps.txt 2>&1
python bench.py > $OUT/r01_${V}_bench.json 2> $OUT/bench.err
tail -c 600 $OUT/r01_${V}_bench.json

# usage: collect_profiles.sh <tag> [landmarks] [full]
#   tag        file prefix under gpurun_out/<tag>/ (e.g. r02_N500)
#   landmarks  filter size (default 200)
#   full       "full" also runs the frame trace / host share / bench.json (N=200 house-keeping)
set -x
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
V=${1:-r02_v0}
NL=${2:-200}
FULL=${3:-}
OUT=$R/gpurun_out/$V
mkdir -p $OUT
ST=50; WU=10
if [ "$NL" -ge 400 ]; then ST=30; WU=5; fi
BCMD="python $R/bench.py --landmarks $NL --steps $ST --warmup $WU --no-cpu-baseline --no-roofline --no-multi-filter --no-frame-mix --no-binding --no-sizes"
rm -rf /tmp/p_kt /tmp/p_f /tmp/p_w /tmp/p_m
rocprofv3 --kernel-trace --stats -d /tmp/p_kt -o kt -- $BCMD > /tmp/kt.log 2>&1
DB=$(find /tmp/p_kt -name "*.db" | head -1)
python $R/scripts/rocpd_stats.py $DB > $OUT/${V}_kernel_stats.csv
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d /tmp/p_f -o f -- $BCMD > /tmp/f.log 2>&1
python $R/scripts/rocpd_pmc.py $(find /tmp/p_f -name "*.db" | head -1) > $OUT/${V}_pmc_fetch_size.csv
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d /tmp/p_w -o w -- $BCMD > /tmp/w.log 2>&1
python $R/scripts/rocpd_pmc.py $(find /tmp/p_w -name "*.db" | head -1) > $OUT/${V}_pmc_write_size.csv
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE -d /tmp/p_m -o m -- $BCMD > /tmp/m.log 2>&1
python $R/scripts/rocpd_pmc.py $(find /tmp/p_m -name "*.db" | head -1) > $OUT/${V}_pmc_mfma.csv
cd $R
python scripts/frame_trace.py $NL 2000 > $OUT/${V}_frame_trace.txt 2>&1
LST=2000; LWU=200; if [ "$NL" -ge 400 ]; then LST=600; LWU=100; fi
python bench.py --landmarks $NL --steps $LST --warmup $LWU --no-multi-filter --no-binding --no-sizes > $OUT/${V}_bench.json 2> $OUT/bench.err
if [ "$FULL" = "full" ]; then
  python scripts/host_share.py $NL 6000 > $OUT/${V}_host_share.txt 2>&1
  python scripts/propagate_vs_steps.py > $OUT/${V}_propagate_vs_steps.txt 2>&1
fi
tail -c 600 $OUT/${V}_bench.json

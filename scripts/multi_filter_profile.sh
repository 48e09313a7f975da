#!/bin/bash
# rocprofv3 kernel trace of R filters on one GPU: gpurun_out/multi_R<R>_overlap.txt. usage: multi_filter_profile.sh "1 3 4 8" [N]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for r in $1; do
  rm -rf /tmp/mf_$r
  rocprofv3 --kernel-trace -d /tmp/mf_$r -o t -- python $R/scripts/multi_filter.py ${2:-200} 400 $r > /tmp/mf_$r.log 2>&1
  grep "filters=" /tmp/mf_$r.log
  python $R/scripts/multi_filter_overlap.py $(find /tmp/mf_$r -name "*.db" | head -1) | tee $R/gpurun_out/multi_R${r}_overlap.txt
done

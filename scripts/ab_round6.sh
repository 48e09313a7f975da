#!/bin/bash
# usage: scripts/ab_round6.sh <tag> <variant> [<variant> ...]   ("tree" = the tree's own libraries): the look-ahead tests on the tree, then same-box alternating bench runs
# and a look-ahead trace per variant; everything under gpurun_out/<tag>/
tag=$1; shift
mkdir -p gpurun_out/$tag
python -m pytest tests/test_gpu_parity.py tests/test_gpu_robustness.py tests/test_gpu_filter_headline.py tests/test_gpu_edge_cases.py -m gpu -x -q -k "lookahead or headline or stalled or concurrent" > gpurun_out/$tag/pytest.log 2>&1
tail -2 gpurun_out/$tag/pytest.log
bash scripts/ab_builds3.sh "$@" > gpurun_out/$tag/ab.txt 2>&1
cat gpurun_out/$tag/ab.txt
for v in "$@"; do
  if [ "$v" = tree ]; then python scripts/lookahead_trace.py 200 > gpurun_out/$tag/trace_tree.txt 2>&1
  else EQVIO_AMD_LIB_DIR=$PWD/$v python scripts/lookahead_trace.py 200 > gpurun_out/$tag/trace_$(basename $v).txt 2>&1; fi
done

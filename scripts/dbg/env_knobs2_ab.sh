#!/bin/bash
# Round 6: HIP runtime knobs that decide what a kernel boundary does to the caches (AMD_OPT_FLUSH: device-scope instead of system-scope fences where the runtime can;
# ROC_SYSTEM_SCOPE_SIGNAL; HIP_HOST_COHERENT; GPU_FLUSH_ON_EXECUTION), same box, one process each, alternating: updates/s at N = $1 (scripts/ab_option.py with a no-op option pair)
N=${1:-50}; F=${2:-600}
run() { env "$@" timeout 120 python scripts/ab_option.py 17:1:1 $N $F 2>/dev/null | head -8 | awk '{s+=$NF==""?0:$(NF-1); n++} END {printf "%.0f", s/n}'; }
for rep in 1; do
  for kv in "X_NONE=1" "AMD_OPT_FLUSH=0" "ROC_SYSTEM_SCOPE_SIGNAL=0" "HIP_HOST_COHERENT=1" "GPU_FLUSH_ON_EXECUTION=1" "X_NONE=2"; do
    echo "N=$N $kv $(run $kv)"
  done
done

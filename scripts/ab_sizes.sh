#!/bin/bash
# usage: scripts/ab_sizes.sh <tag> "<sizes>" <variant> [...]: same-box alternating bench lines at several landmark counts
tag=$1; shift; sizes=$1; shift
mkdir -p gpurun_out/$tag
for N in $sizes; do N=$N bash scripts/ab_builds3.sh "$@" 2>&1 | sed "s/^/N=$N /"; done | tee gpurun_out/$tag/ab_sizes.txt

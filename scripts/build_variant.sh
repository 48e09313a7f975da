#!/bin/bash
# Build a variant of the libraries into scripts/ab_libs_<name>/ (same-box A/B with scripts/ab_builds3.sh): scripts/build_variant.sh <name> [-DEQF_...=.. ...]
set -e
name=$1; shift
out=scripts/ab_libs_$name
mkdir -p $out
( cd eqvio_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -Wall -Wno-unused "$@" -shared -o ../../$out/libeqf_hip.so eqf_hip.hip 2>&1 | grep -E "error|spill" || true )
cp eqvio_amd/lib/libeqvio_filter.so $out/
ls -la $out/libeqf_hip.so

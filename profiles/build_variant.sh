#!/bin/bash
# Builds a variant of librender_mi355x.so for A/B timing:  bash profiles/build_variant.sh <name> "<extra hipcc flags>"
# -> profiles/microbench/lib_<name>.so (git-ignored; travels to the GPU box).  Use with APT_LIB_PATH=...
set -e
cd "$(dirname "$0")/../ascendpathtracing_amd/csrc"
mkdir -p ../../profiles/microbench
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -mllvm -enable-post-misched=false -mllvm -disable-vector-combine $2 -Wno-unused-function -c render_kernels.hip -o /tmp/rk_$1.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../profiles/microbench/lib_$1.so /tmp/rk_$1.o host_helpers.o render_do_cxx.o
echo built profiles/microbench/lib_$1.so

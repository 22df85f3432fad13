#!/bin/bash
# Builds a variant of librender_mi355x.so for A/B timing or tests:  bash profiles/build_variant.sh <name> "<extra hipcc flags>"
# -> profiles/microbench/lib_<name>.so (git-ignored; travels to the GPU box).  Use with APT_LIB_PATH=...
# Same flags, export list and host objects as the product build (csrc/Makefile target `variant`).
set -e
cd "$(dirname "$0")/../ascendpathtracing_amd/csrc"
make -s host_helpers.o render_do_cxx.o
make -s variant NAME="$1" EXTRA="$2"
echo built profiles/microbench/lib_$1.so

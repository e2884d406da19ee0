#!/bin/bash
# Builds a variant of libfrieda_hip.so with extra compiler flags into build_exp/<name>/ (for A/B runs on the GPU box:
# FRIEDA_HIP_LIB=build_exp/<name>/libfrieda_hip.so python bench.py ...).  usage: tools/build_variant.sh <name> [flags...]
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build_exp/$name
make -C $root/frieda_amd/csrc -s -j8 OUT=$root/build_exp/$name OBJ=$root/build_exp/$name/obj \
  CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wextra -Wno-unused-parameter -ffp-contract=off $*"
ls -la $root/build_exp/$name/libfrieda_hip.so

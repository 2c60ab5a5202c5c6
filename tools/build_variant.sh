#!/bin/bash
# Build an experimental variant of the product library with extra compiler flags into build/exp/<name>.so (same ABI; loaded by
# bench.py / the tools through LSLAM_LIB=...): tools/build_variant.sh <name> "<extra flags>"
set -e
name=$1; shift
root=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p $root/build/exp
make -s -C $root/the-cooper-mapper_amd/csrc -j8 OBJDIR=$root/build/exp/obj_$name OUT=$root/build/exp/$name.so EXTRA="$*" > /dev/null
ls -la $root/build/exp/$name.so

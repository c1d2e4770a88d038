#!/bin/bash
# tools/build_variant.sh NAME FILE.hip "-DFLAG=..."  -> echoseal_amd/libechoseal_hip_NAME.so
# A/B builds of one translation unit: compiles FILE.hip with the extra flags and links it with the other (already built) objects.
set -e
cd "$(dirname "$0")/../echoseal_amd/csrc"
name=$1; file=$2; flags=$3
make -s >/dev/null
obj=/tmp/variant_${name}_$(basename ${file%.hip}).o
per_file=$(sed -n "s/^FLAGS_$(basename ${file%.hip}) := //p" Makefile)      # the Makefile's per-file flags (es_sync32: no SLP vectorisation)
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function $per_file $flags -c $file -o $obj
others=$(ls *.o | grep -v "^$(basename ${file%.hip}).o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../libechoseal_hip_${name}.so $obj $others
echo built ../libechoseal_hip_${name}.so

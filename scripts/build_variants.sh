#!/bin/bash
# Build A/B variants of libgoat_hip.so: scripts/build_variants.sh name1:"-DX=1 -DY=2" name2:"..."  -> vln-goat_amd/csrc/ab/libgoat_<name>.so
# (only gemm2.hip / gemm3.hip are recompiled with the extra flags; the other objects come from the regular build)
set -e
cd "$(dirname "$0")/../vln-goat_amd/csrc"
mkdir -p ab
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  (
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c gemm2.hip -o ab/gemm2_$name.o &
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c gemm3.hip -o ab/gemm3_$name.o &
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libgoat_$name.so ab/gemm2_$name.o ab/gemm3_$name.o $(ls *.o | grep -v '^gemm[23]\.o$')
    echo built ab/libgoat_$name.so
  ) &
done
wait

#!/bin/bash
# Build A/B variants of libgoat_hip.so: scripts/build_variants.sh name1:"-DX=1 -DY=2" name2:"..."  -> vln-goat_amd/csrc/ab/libgoat_<name>.so
# (only gemm2.hip / gemm3.hip — or the sources named in VARIANT_SRC, e.g. VARIANT_SRC="rowops" — are recompiled with the extra
#  flags; the other objects come from the regular build)
set -e
cd "$(dirname "$0")/../vln-goat_amd/csrc"
mkdir -p ab
for spec in "$@"; do
  name="${spec%%:*}"; flags="${spec#*:}"
  (
    objs=""; skip=""
    for src in ${VARIANT_SRC:-gemm2 gemm3}; do
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC $flags -c $src.hip -o ab/${src}_$name.o &
      objs="$objs ab/${src}_$name.o"; skip="$skip -e ^$src\.o\$"
    done
    wait
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ab/libgoat_$name.so $objs $(ls *.o | grep -v $skip)
    echo built ab/libgoat_$name.so
  ) &
done
wait

#!/bin/bash
# A/B builds of the rasterizer library with extra -D flags, timed back to back on ONE lease (tools/quick_timing.py through DGS_SURFEL_LIB).
#   here:        tools/ab_variants.sh build  name1:-DFOO=1 name2:"-DBAR=2 -DBAZ" ...     (writes gpurun_out/ab/lib_<name>.so ... no: csrc/ab_<name>.so, travels with the snapshot)
#   on the box:  tools/ab_variants.sh run [quick_timing args]
R=$(cd "$(dirname "$0")/.." && pwd)
D=$R/dynamic-2dgs_amd/csrc
if [ "$1" = build ]; then
  shift
  rm -f $D/ab_*.so
  for v in "$@"; do
    name=${v%%:*}; flags=${v#*:}
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -fno-slp-vectorize -DDGS_AB_BUILD $flags $D/surfel_rasterizer.hip -o $D/ab_$name.so 2>&1 | grep -v warning | grep -v "^$" &
  done
  wait
  ls $D/ab_*.so
else
  shift
  for so in $D/ab_*.so; do
    printf "%-28s " $(basename $so .so)
    DGS_SURFEL_LIB=$so python $R/tools/quick_timing.py "$@" 2>&1 | tail -2 | tr '\n' ' '
    echo
  done
fi

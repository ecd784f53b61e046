#!/bin/bash
# Builds variants of the library that differ only in the generated attention loop (tools/gen_attn_asm.py):
#   tools/build_attn_variants.sh name1:"A64A_ABL=noexp" name2:"A64A_OPT=x=1" ...   -> pi3_slam_amd/libpi3slam_hip_v<name>.so
# for side-by-side timing with tools/dev_attn_ab.py.  The committed attn64a_loop.inc is not touched.
set -e
cd "$(dirname "$0")/../pi3_slam_amd/csrc"
mkdir -p build_abl
OBJS=$(ls build/*.o | grep -v attn64.o | tr '\n' ' ')
for spec in "$@"; do
  name=${spec%%:*}; envs=${spec#*:}
  (
    d=build_abl/v_$name; mkdir -p $d
    env $envs A64A_OUT=$PWD/$d/attn64a_loop.inc python ../../tools/gen_attn_asm.py > /dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -Wno-unused-result -Wno-inline-asm -fno-slp-vectorize \
        -I$d -I../../include -I. -c attn64.hip -o $d/attn64.o 2> $d/err.log || { cat $d/err.log | grep error -A3; exit 1; }
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libpi3slam_hip_v$name.so $OBJS $d/attn64.o
    echo built $name
  ) &
done
wait

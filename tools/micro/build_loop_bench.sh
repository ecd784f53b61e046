#!/bin/bash
# tools/micro/build_loop_bench.sh name:ABL ...  -> tools/micro/attn_loop_bench_<name> (both NBLK variants with that ablation set)
cd "$(dirname "$0")/../.."
for spec in "$@"; do
  name=${spec%%:*}; abl=${spec#*:}
  ( d=/tmp/alb_$name; mkdir -p $d
    A64A_ABL=$abl python tools/gen_attn_asm2.py 2 $d/a64b2.inc > /dev/null && A64A_ABL=$abl python tools/gen_attn_asm2.py 4 $d/a64b4.inc > /dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -Wno-inline-asm -I pi3_slam_amd/csrc -I $d -o tools/micro/attn_loop_bench_$name tools/micro/attn_loop_bench.hip 2>&1 | grep -E "error" -A3
    echo built $name ) &
done
wait

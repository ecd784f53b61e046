#!/bin/bash
# online stream: graph / eager x 4 / 8 hardware queues, producer-thread pipeline
mkdir -p gpurun_out/gr
for hq in 4 8; do for g in 1 0; do
  GPU_MAX_HW_QUEUES=$hq PI3_DEV_GRAPH=$g timeout -k 10 300 python tools/dev_online_stream.py 1200 > gpurun_out/gr/t_${hq}_${g}.log 2>&1 || { tail -20 gpurun_out/gr/t_${hq}_${g}.log; exit 1; }
  echo "HWQ=$hq $(grep RESULT gpurun_out/gr/t_${hq}_${g}.log)"
done; done

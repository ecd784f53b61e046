#!/bin/bash
# kernel trace of the online stream (600 frames) in hipGraph mode and with plain launches: gaps on the compute queue
mkdir -p gpurun_out/gr2
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for g in 1 0; do
  PI3_DEV_GRAPH=$g timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/gr2/prof_$g -o t -- python3 $R/tools/dev_online_stream.py 600 > $R/gpurun_out/gr2/p$g.log 2>&1
  grep RESULT $R/gpurun_out/gr2/p$g.log
done
cd $R
python3 - <<'PY'
import csv, glob
for g in ("1", "0"):
    f = glob.glob(f"gpurun_out/gr2/prof_{g}/**/*kernel_trace.csv", recursive=True)[0]
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
    rows.sort(key=lambda r: r["s"])
    qmain = max(set(r["Queue_Id"] for r in rows), key=lambda q: sum(1 for r in rows if r["Queue_Id"] == q))
    tend = rows[-1]["e"]
    q1 = [r for r in rows if r["Queue_Id"] == qmain and r["s"] > tend - 2.3e9]
    cur = q1[0]["e"]; gaps = []
    for a, b in zip(q1, q1[1:]):
        gap = (b["s"] - cur) / 1e6
        if gap > 0.05: gaps.append((round(gap, 2), a["Kernel_Name"][:28], b["Kernel_Name"][:28]))
        cur = max(cur, b["e"])
    busy = sum(r["e"] - r["s"] for r in q1) / 1e6
    print("graph" if g == "1" else "eager", "main-queue kernels", len(q1), "busy ms", round(busy, 1), "span ms", round((q1[-1]["e"] - q1[0]["s"]) / 1e6, 1))
    print("   gaps > 0.05 ms:", sorted(gaps, reverse=True)[:12], "sum", round(sum(x[0] for x in gaps), 1))
PY
find gpurun_out/gr2 -name "*kernel_trace.csv" -delete

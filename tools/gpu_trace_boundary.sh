#!/bin/bash
# what runs between the last kernel of one chunk's forward and the first of the next (bench command, eager launches)
mkdir -p gpurun_out/gr3
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/gr3/prof -o t -- python3 $R/bench.py --steps 4 --warmup 2 --no-extras --no-cpu-baseline > $R/gpurun_out/gr3/b.json 2> $R/gpurun_out/gr3/b.err
cd $R
python3 - <<'PY'
import csv, glob
f = glob.glob("gpurun_out/gr3/prof/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r["s"] = int(r["Start_Timestamp"]); r["e"] = int(r["End_Timestamp"])
rows.sort(key=lambda r: r["s"])
unp = [r for r in rows if "unpatchify_points" in r["Kernel_Name"]]
pg = [r for r in rows if "patch_gather" in r["Kernel_Name"]]
u = unp[-2]
nxt = [r for r in pg if r["s"] > u["e"]][0]
print("from unpatchify end to next patch_gather start: %.2f ms" % ((nxt["s"] - u["e"]) / 1e6))
for r in rows:
    if r["e"] >= u["s"] and r["s"] <= nxt["e"]:
        print("  q%s  +%8.3f ms  dur %7.3f ms  %s" % (r["Queue_Id"], (r["s"] - u["e"]) / 1e6, (r["e"] - r["s"]) / 1e6, r["Kernel_Name"][:70]))
mc = glob.glob("gpurun_out/gr3/prof/**/*memory_copy_trace.csv", recursive=True)
if mc:
    for r in csv.DictReader(open(mc[0])):
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        if e >= u["s"] and s <= nxt["e"]:
            print("  copy +%8.3f ms dur %7.3f ms %s" % ((s - u["e"]) / 1e6, (e - s) / 1e6, r.get("Direction", "")))
PY
find gpurun_out/gr3 -name "*trace.csv" -delete

#!/bin/bash
# Run ON THE GPU BOX: kernel timeline of the mono flow loop (which stream is the long one?)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/ftrace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/ftrace -- python3 $R/tools/flow_loop.py 30 256 > $R/gpurun_out/ftrace.log 2>&1
tail -1 $R/gpurun_out/ftrace.log
python3 - <<'PY'
import csv, glob, os, collections
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/ftrace/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
mk = [r for r in rows if "match_kernel" in r["Kernel_Name"]]
t0 = int(mk[-12]["Start_Timestamp"])
print("last steps, per queue (ms from the 12th-last search):")
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if s < t0 or s > int(mk[-8]["End_Timestamp"]):
        continue
    n = r["Kernel_Name"].split("::")[-1].split("(")[0][:28]
    print(f"  q{r['Queue_Id']:>2} {1e-6*(s-t0):8.3f} -> {1e-6*(e-t0):8.3f}  {1e-3*(e-s):8.1f} us  {n}")
PY

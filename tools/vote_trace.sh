#!/bin/bash
# Run ON THE GPU BOX: kernel timeline of the device e2e loop (do the vote batches overlap each other and the matcher?)
#   tools/vote_trace.sh B NB LANES [STEPS]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/vtrace
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/vtrace -- python3 $R/bench.py --no-cpu --no-other --no-exclusive --no-e2e-host --steps 5 --warmup 2 --blocks 1 --e2e-steps-per-batch $1 --e2e-batches $2 --e2e-lanes $3 --e2e-steps ${4:-48} > $R/gpurun_out/vtrace.json 2> $R/gpurun_out/vtrace.err
python3 - <<'PY'
import csv, glob, os
R = os.environ["GRAFT_REPO_ROOT"]
f = glob.glob(R + "/gpurun_out/vtrace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
sw = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id"), r.get("Stream_Id", "")) for r in rows if "vote_sweep" in r["Kernel_Name"]]
mk = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows if "match_kernel" in r["Kernel_Name"]]
t0 = min(s[0] for s in sw)
print("vote_sweep launches:")
for s in sw:
    inside = sum(1 for m in mk if m[0] >= s[0] and m[1] <= s[1])
    print(f"  start {1e-6*(s[0]-t0):9.2f} ms  dur {1e-6*(s[1]-s[0]):8.2f} ms  queue {s[2]} stream {s[3]}  match kernels inside: {inside}")
md = sorted(1e-3 * (m[1] - m[0]) for m in mk)
print(f"match_kernel: {len(mk)} launches, median {md[len(md)//2]:.0f} us")
names = {}
for r in rows:
    names.setdefault(r["Kernel_Name"][:60], []).append(1e-3 * (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in sorted(names.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print(f"  {k:60s} n {len(v):5d}  total {sum(v)/1e3:9.2f} ms  mean {sum(v)/len(v):9.1f} us")
PY

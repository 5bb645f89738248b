#!/bin/bash
# Run ON THE GPU BOX (through gpurun): per-dispatch durations of the mono estimator's kernels (tools/mono_timing.py S lists)
#   tools/mono_kernels.sh [S]  ->  gpurun_out/mono_kernels.txt
S=${1:-64}
cd /tmp && export TMPDIR=/tmp
rm -rf $GRAFT_REPO_ROOT/gpurun_out/prof_mono
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_mono -- python $GRAFT_REPO_ROOT/tools/mono_timing.py $S > $GRAFT_REPO_ROOT/gpurun_out/prof_mono.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY' > gpurun_out/mono_kernels.txt
import csv, glob, collections
f = glob.glob('gpurun_out/prof_mono/**/*kernel_trace.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if 'mono' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(n): return 'final_a' if 'final_a' in n else 'final_c' if 'final_c' in n else 'tri' if 'mono_tri' in n else 'norm' if 'norm' in n else 'hyp' if '<false>' in n else 'hyp_signed'
calls, cur = [], {}
for r in rows:
    k = short(r['Kernel_Name'])
    if k == 'norm' and cur: calls.append(cur); cur = {}
    cur[k] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
calls.append(cur)
print("per call, us (phases of tools/mono_timing.py: 4 calls each of 400 matches, 2000 matches, unbucketed group):")
for c in calls: print("  " + "  ".join(f"{k} {v:8.1f}" for k, v in c.items()) + f"   sum {sum(c.values()):8.1f}")
PY
cat gpurun_out/mono_kernels.txt

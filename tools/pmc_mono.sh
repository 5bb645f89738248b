#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ counter passes over the mono estimator's kernels (tools/mono_timing.py, 64 lists)
#   tools/pmc_mono.sh  ->  gpurun_out/pmc_mono/p<i>/..., gpurun_out/pmc_mono.txt.   PMC passes only (no trace domains mixed in).
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_mono
rm -rf $OUT && mkdir -p $OUT
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_IFETCH" \
           "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_SALU" \
           "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA"; do
  i=$((i+1))
  timeout -k 10 240 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python tools/mono_timing.py 64 > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; echo "pass $i failed: $set"; continue; }
  echo "pass $i ok: $set"
done
python - <<'PY' > gpurun_out/pmc_mono.txt
import csv, glob, collections
tot = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for f in glob.glob('gpurun_out/pmc_mono/p*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        n = r['Kernel_Name']
        if 'mono' not in n: continue
        k = 'final_a' if 'final_a' in n else 'final_c' if 'final_c' in n else 'tri' if 'mono_tri' in n else 'norm' if 'norm' in n else 'hyp' if '<false>' in n else 'hyp_signed'
        tot[k][r['Counter_Name']] += float(r['Counter_Value'])
for k, d in tot.items():
    print(k, {c: f"{v:.4g}" for c, v in sorted(d.items())})
PY
cat gpurun_out/pmc_mono.txt

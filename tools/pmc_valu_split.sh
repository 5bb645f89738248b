#!/bin/bash
# Run ON THE GPU BOX: SQ_INSTS_VALU of match_kernel for the shipped library and for the timing-only builds without the
# stereo / without the flow passes (make VARIANT=nostereo EXTRA=-DVH_EXP_SKIP=1, VARIANT=noflow EXTRA=-DVH_EXP_SKIP=2):
# where the search kernel's instructions go (profiles/EXPERIMENTS.md, round 5).  S = 128, exclusive kernels.
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
export VH_SERIAL=1
for v in "" _nostereo _noflow; do
  OUT=gpurun_out/pmc_split$v
  rm -rf $OUT
  VISO_HIP_LIB=$PWD/hls-final-visual-odometry_amd/libviso_hip$v.so rocprofv3 --pmc SQ_INSTS_VALU --output-format csv -d $OUT -- python bench.py --no-cpu --no-other --no-exclusive --no-e2e --no-flow --streams 128 --steps 4 --warmup 2 --blocks 1 > $OUT.log 2>&1 || { tail -3 $OUT.log; continue; }
  python - "$OUT" "${v:-shipped}" <<'PY'
import csv, glob, sys, os
f = max(glob.glob(os.path.join(sys.argv[1], "*", "*counter_collection.csv")), key=os.path.getmtime)
v = sorted(float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "match_kernel" in r["Kernel_Name"])
print(f"{sys.argv[2]:10s} match_kernel SQ_INSTS_VALU per S=128 launch (median of {len(v)}): {v[len(v)//2]:.4e} = {v[len(v)//2]/128/1e6:.3f} M per stereo pair")
PY
done

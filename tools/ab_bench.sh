#!/bin/bash
# Run ON THE GPU BOX (through gpurun): A/B of experiment libraries (make VARIANT=name EXTRA=...).
#   tools/ab_bench.sh "<variant> [bench args]" ...      (variant "-" = the shipped library)
# One bench.py run per argument; prints value, match exclusive / overlapped us, re-search share.
cd "$GRAFT_REPO_ROOT"
i=0
for spec in "$@"; do
  i=$((i+1))
  set -- $spec
  v=$1; shift
  lib=hls-final-visual-odometry_amd/libviso_hip.so
  [ "$v" != "-" ] && lib=hls-final-visual-odometry_amd/libviso_hip_$v.so
  out=gpurun_out/ab_${i}_${v}.json
  VISO_HIP_LIB=$PWD/$lib python bench.py --no-cpu --no-other "$@" > $out 2> ${out%.json}.err || { echo "$spec: FAILED"; tail -3 ${out%.json}.err; continue; }
  python - "$out" "$spec" <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
k, x = d["kernels_us_per_launch"], d["kernels_us_per_launch_exclusive"]
print(f"{sys.argv[2]:40s} {d['value']:9.0f} pairs/s  match {k.get('match', 0):7.1f} (excl {x.get('match', 0):7.1f})  chain {k.get('chain', 0):6.1f} (excl {x.get('chain', 0):6.1f})  "
      f"redo {d['search_loop']['queries_searched_again']:.4f} {d['search_loop']['form']}", flush=True)
PY
done

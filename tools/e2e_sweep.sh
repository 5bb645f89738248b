#!/bin/bash
# Run ON THE GPU BOX: the device form of the as-shipped loop (bench.py: e2e_matchfeatures) for several shapes of the
# vote pipeline.   tools/e2e_sweep.sh "B NB LANES [PRIO]" ...
cd "$GRAFT_REPO_ROOT"
i=0
for spec in "$@"; do
  i=$((i+1))
  set -- $spec
  out=gpurun_out/e2e_${i}.json
  VH_VOTE_PRIO=${4:-0} timeout -k 10 240 python bench.py --no-cpu --no-other --no-exclusive --no-e2e-host --e2e-steps-per-batch $1 --e2e-batches $2 --e2e-lanes $3 ${5:+--e2e-steps $5} > $out 2> ${out%.json}.err || { echo "$spec: FAILED"; tail -3 ${out%.json}.err; continue; }
  python - "$out" "$spec" <<'PY'
import json, sys
s = open(sys.argv[1]).read()
d = json.loads(s[s.index('{"metric'):])
e = d["e2e_matchfeatures"]
print(f"{sys.argv[2]:16s} headline {d['value']:8.0f}  e2e {e['value']:8.0f} pairs/s  {e['ms_per_step']:6.2f} ms/step  steps {e['steps']} in flight {e['steps_in_flight']}  finish-wait {e['host_ms_per_step_waiting_in_finish']:.2f} ms  ok {e['pose_ok_share']:.2f}", flush=True)
PY
done

#!/bin/bash
# Run ON THE GPU BOX: HBM traffic of the device vote's kernels on an idle chip (rocprofv3 FETCH_SIZE / WRITE_SIZE / L2 hit
# counters, separate passes) -> gpurun_out/<tag>_vote_traffic.txt.   tools/vote_traffic.sh <tag> [lists] [lanes]
TAG=${1:-r05}; LISTS=${2:-8192}; LANES=${3:-64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  d=gpurun_out/vtraf_$(echo $c | cut -d' ' -f1)
  rm -rf $d
  rocprofv3 --pmc $c --output-format csv -d $d -- python tools/vote_timing.py --lists $LISTS --lanes $LANES > $d.log 2>&1 || tail -3 $d.log
done
python - "$TAG" "$LISTS" "$LANES" <<'PY'
import csv, glob, os, re, sys, collections
tag, lists, lanes = sys.argv[1], int(sys.argv[2]), sys.argv[3]
val = collections.defaultdict(dict)
for d in glob.glob("gpurun_out/vtraf_*"):
    if not os.path.isdir(d):
        continue
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    seen = set()
    for r in csv.DictReader(open(f)):
        m_ = re.search(r"(\w+_kernel)", r["Kernel_Name"]); k = m_.group(1) if m_ else r["Kernel_Name"]
        if (k, r["Counter_Name"]) in seen:  # first launch of each kernel
            continue
        seen.add((k, r["Counter_Name"]))
        val[k][r["Counter_Name"]] = float(r["Counter_Value"])
out = [f"# rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum (three passes) -- python tools/vote_timing.py --lists {lists} --lanes {lanes}",
       "# idle chip; raw counter values (KiB), first launch of each kernel"]
for k, v in val.items():
    if not k.startswith("vote_"):
        continue
    fe, wr = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
    out.append(f"{k:22s} fetch {fe / 1024:9.1f} MiB  write {wr / 1024:9.1f} MiB  per list: fetch {fe / lists:8.1f} KiB write {wr / lists:8.1f} KiB   "
               f"L2 requests {v.get('TCC_REQ_sum', 0):.3g} hits {v.get('TCC_HIT_sum', 0):.3g} misses {v.get('TCC_MISS_sum', 0):.3g}")
open(f"gpurun_out/{tag}_vote_traffic.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY

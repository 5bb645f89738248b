#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU
box) into the tracked evidence under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary of `python bench.py --no-cpu`
  profiles/<tag>_traffic.json       per-kernel HBM bytes per launch from the FETCH_SIZE / WRITE_SIZE passes
  profiles/<tag>_bench.json         the plain `python bench.py` line of the same build
  profiles/traffic_latest.json      what bench.py reads for roofline.traffic

gfx950 counter handling (MI355X_MICROARCH.md, HBM section): FETCH_SIZE and
WRITE_SIZE are in KiB; FETCH_SIZE tallies 128-B requests at 64 B for wide
(16 B/lane) coalesced reads, so it is doubled; WRITE_SIZE is taken as is.
Kernels here mix 16-B and narrower accesses, so the corrected figure is an
upper estimate of the read side; the raw counters are kept next to it.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "prof_" + tag)
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)

NAMES = {"detect_nms_fast_kernel": "detect_nms", "detect_nms_kernel": "detect_nms", "emit_features_kernel": "emit_features",
         "bin_scan_kernel": "bin_scan", "bin_sort_kernel": "bin_sort", "bin_hist_kernel": "bin_hist", "bin_fill_kernel": "bin_fill",
         "match_kernel": "match", "chain_kernel": "chain",
         "emit_matches_kernel": "emit_matches", "flow_keep_kernel": "flow_keep"}


def short(name):
    m = re.search(r"::(\w+)[<(]", name)
    k = m.group(1) if m else name
    return NAMES.get(k, k)


def newest(pattern):
    return max(glob.glob(pattern), key=os.path.getmtime)


stats = newest(os.path.join(src, "stats", "*", "*kernel_stats.csv"))
shutil.copy(stats, os.path.join(dst, tag + "_kernel_stats.csv"))
avg_us = {short(r["Name"]): float(r["AverageNs"]) / 1e3 for r in csv.DictReader(open(stats))}

per = collections.defaultdict(dict)
for kind in ("fetch", "write"):
    f = newest(os.path.join(src, kind, "*", "*counter_collection.csv"))
    acc, cnt = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(f)):
        k = short(r["Kernel_Name"])
        acc[k] += float(r["Counter_Value"])
        cnt[k] += 1
    for k in acc:
        per[k][kind + "_size_kib_per_launch"] = acc[k] / cnt[k]

bench = json.loads(open(os.path.join(src, "bench_plain.json")).read().strip().splitlines()[-1])
S = bench["config"]["streams_per_gpu"]
out = {"tag": tag, "streams": S,
       "command": "python bench.py --no-cpu (rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE, separate passes)", "kernels": {}}
for k, v in per.items():
    if "rocclr" in k:
        continue
    fe, wr = v.get("fetch_size_kib_per_launch", 0.0), v.get("write_size_kib_per_launch", 0.0)
    out["kernels"][k] = {**v, "hbm_bytes_per_launch_raw": (fe + wr) * 1024, "hbm_bytes_per_launch": (2 * fe + wr) * 1024,
                         "avg_us_rocprof_stats": avg_us.get(k)}
json.dump(out, open(os.path.join(dst, tag + "_traffic.json"), "w"), indent=1)
json.dump(bench, open(os.path.join(dst, tag + "_bench.json"), "w"), indent=1)
dom = bench["roofline"]["kernel"]
# what bench.py reads: tagged with a hash of the SOURCES that determine the code object (the driver rebuilds the
# library, and every rebuild of the .so hashes differently: round 4's line lost its traffic to that)
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402
json.dump({"tag": tag, "streams": S, "kernel": dom, "src_sha256": entry.load_package().source_sha256(),
           "hbm_bytes_per_launch": out["kernels"][dom]["hbm_bytes_per_launch"],
           "per_kernel": {k: v["hbm_bytes_per_launch"] for k, v in out["kernels"].items()},
           "per_kernel_raw": {k: v["hbm_bytes_per_launch_raw"] for k, v in out["kernels"].items()},
           "note": "bytes per LAUNCH from rocprofv3 FETCH_SIZE / WRITE_SIZE (separate passes); per_kernel = 2 x FETCH + WRITE "
                   "(the guide's gfx950 correction for wide reads: an upper estimate here), per_kernel_raw = FETCH + WRITE"},
          open(os.path.join(dst, "traffic_latest.json"), "w"), indent=1)
print("value", bench["value"], "dominant", dom, "us(events)", bench["roofline"]["us_per_launch"], "us(rocprof)", avg_us.get(dom))
for k, v in sorted(out["kernels"].items(), key=lambda kv: -(kv[1]["avg_us_rocprof_stats"] or 0)):
    print(f"{k:14s} {v['avg_us_rocprof_stats'] or 0:9.1f} us  fetch {v.get('fetch_size_kib_per_launch', 0) / 1024:8.1f} MiB"
          f"  write {v.get('write_size_kib_per_launch', 0) / 1024:8.1f} MiB")

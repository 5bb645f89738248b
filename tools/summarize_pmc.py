#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/p*/ (tools/pmc_kernels.sh) -> profiles/<tag>_pmc.json:
SQ counters per kernel and launch (exclusive kernels, VH_SERIAL=1, S = 128)."""
import collections, csv, glob, json, os, re, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out", "pmc_" + tag)
vals = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sorted(glob.glob(os.path.join(src, "p*"))):
    if not os.path.isdir(d):
        continue
    # gpurun merges into an existing directory: an earlier run's files may lie beside this run's
    f = max(glob.glob(os.path.join(d, "*", "*counter_collection.csv")), key=os.path.getmtime)
    for r in csv.DictReader(open(f)):
        m = re.search(r"::(\w+)[<(]", r["Kernel_Name"])
        k = m.group(1) if m else r["Kernel_Name"]
        vals[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {"note": "rocprofv3 --pmc SQ_* on `VH_SERIAL=1 python bench.py --streams 128 --no-cpu` (exclusive kernels); "
               "values are the MEDIAN over the launches of a kernel, i.e. per full step of 128 stereo pairs "
               "(the first match of a run, against an empty previous frame, is far cheaper and would drag a mean down)", "kernels": {}}
for k in vals:
    if "rocclr" in k:
        continue
    out["kernels"][k] = {c: sorted(vals[k][c])[len(vals[k][c]) // 2] for c in sorted(vals[k])}
json.dump(out, open(os.path.join(ROOT, "profiles", tag + "_pmc.json"), "w"), indent=1)
for k, v in out["kernels"].items():
    g = lambda n: v.get(n, 0.0)
    if not g("SQ_INSTS_VALU"):
        continue
    print(f"{k:28s} VALU instr {g('SQ_INSTS_VALU'):.3e}  active_valu/busy_cycles {g('SQ_ACTIVE_INST_VALU') / max(g('SQ_BUSY_CYCLES'), 1):.2f}"
          f"  LDS instr {g('SQ_INSTS_LDS'):.3e} lds_idx_active {g('SQ_LDS_IDX_ACTIVE'):.3e} conflicts {g('SQ_LDS_BANK_CONFLICT'):.2e}"
          f"  wave_cycles {g('SQ_WAVE_CYCLES'):.3e} wait_lds {g('SQ_WAIT_INST_LDS'):.3e} wait_any {g('SQ_WAIT_INST_ANY'):.3e}")

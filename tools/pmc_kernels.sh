#!/bin/bash
# Run ON THE GPU BOX (through gpurun): SQ counter passes over exclusive kernels
# (VH_SERIAL=1, S=128) -> gpurun_out/pmc_<tag>/<pass>/...; summarise with
# tools/summarize_pmc.py <tag>.  PMC passes only (no trace domains mixed in).
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/pmc_$TAG
rm -rf $OUT && mkdir -p $OUT
export VH_SERIAL=1
i=0
for set in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES" \
           "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT" \
           "SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VMEM" \
           "SQ_BUSY_CU_CYCLES SQ_WAVES SQ_LEVEL_WAVES SQ_CYCLES"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python bench.py --no-cpu --no-other --no-exclusive --no-e2e --no-flow --streams 128 --steps 4 --warmup 2 --blocks 1 > $OUT/p$i.log 2>&1 || { tail -5 $OUT/p$i.log; exit 1; }
  echo "pass $i ok: $set"
done

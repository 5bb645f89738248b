#!/bin/bash
# Run ON THE GPU BOX (through gpurun): rocprofv3 evidence for one round.
#   tools/profile_round.sh <tag>      ->  gpurun_out/prof_<tag>/{stats,fetch,write}/...
# --kernel-trace/--stats and the PMC passes are separate runs (the pool refuses
# mixing them); FETCH_SIZE and WRITE_SIZE need separate passes (TCC slots).
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/prof_$TAG
rm -rf $OUT
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python bench.py --no-cpu --no-other --no-exclusive --no-e2e --no-flow > $OUT/bench_stats.log 2>&1
tail -1 $OUT/bench_stats.log | cut -c1-300
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -- python bench.py --no-cpu --no-other --no-exclusive --no-e2e --no-flow --steps 6 --warmup 2 --blocks 1 > $OUT/bench_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -- python bench.py --no-cpu --no-other --no-exclusive --no-e2e --no-flow --steps 6 --warmup 2 --blocks 1 > $OUT/bench_write.log 2>&1
# a short plain run names the stream count and the dominant kernel; the summary then writes profiles/traffic_latest.json for
# THIS build (sha256 of the library), so that the full plain run after it carries roofline.traffic
python bench.py --no-cpu --no-other --no-e2e --blocks 1 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
python tools/summarize_profiles.py $TAG > $OUT/summary_on_box.txt 2>&1 || tail -3 $OUT/summary_on_box.txt
python bench.py > $OUT/bench_plain.json 2> $OUT/bench_plain.err
tail -c 600 $OUT/bench_plain.json

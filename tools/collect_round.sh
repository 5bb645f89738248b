#!/bin/bash
# Run ON THE GPU BOX (through gpurun): the tracked evidence of one round in one call.
#   tools/collect_round.sh <tag>   ->  gpurun_out/{prof_<tag>, pmc_<tag>, <tag>_*.json/txt}
TAG=${1:-r04}
cd "$GRAFT_REPO_ROOT"
tools/profile_round.sh $TAG > gpurun_out/${TAG}_profile.log 2>&1 || tail -5 gpurun_out/${TAG}_profile.log
echo "profile_round done"
tools/pmc_kernels.sh $TAG > gpurun_out/${TAG}_pmc.log 2>&1 || tail -5 gpurun_out/${TAG}_pmc.log
echo "pmc done"
for w in 1080p 4k; do python bench.py --workload $w --cpu-seconds 4 > gpurun_out/${TAG}_ctx_$w.json 2> gpurun_out/${TAG}_ctx_$w.err || tail -3 gpurun_out/${TAG}_ctx_$w.err; echo "ctx $w done"; done
for n in 1 2; do python bench.py --noise $n --cpu-seconds 4 --no-e2e > gpurun_out/${TAG}_ctx_noise$n.json 2> gpurun_out/${TAG}_ctx_noise$n.err || tail -3 gpurun_out/${TAG}_ctx_noise$n.err; echo "ctx noise $n done"; done
python tools/mono_timing.py > gpurun_out/${TAG}_mono_timing.txt 2>&1; tail -3 gpurun_out/${TAG}_mono_timing.txt
python tools/ego_timing.py > gpurun_out/${TAG}_ego_timing.txt 2>&1; tail -2 gpurun_out/${TAG}_ego_timing.txt
python tools/latency_one.py > gpurun_out/${TAG}_latency_one.txt 2>&1; tail -1 gpurun_out/${TAG}_latency_one.txt
for fr in 0.05 0.2; do python bench.py --noise 1 --noise-frac $fr --cpu-seconds 4 --no-e2e > gpurun_out/${TAG}_ctx_noise1_frac$fr.json 2> gpurun_out/${TAG}_ctx_noise1_frac$fr.err || tail -3 gpurun_out/${TAG}_ctx_noise1_frac$fr.err; echo "ctx noise 1 frac $fr done"; done
tools/mono_kernels.sh 256 > /dev/null 2>&1; cp gpurun_out/mono_kernels.txt gpurun_out/${TAG}_mono_kernels.txt; tail -12 gpurun_out/${TAG}_mono_kernels.txt
timeout 200 python tools/fuzz_vote.py --seconds 90 --seed 55 > gpurun_out/${TAG}_fuzz_vote.txt 2>&1; tail -2 gpurun_out/${TAG}_fuzz_vote.txt

#!/bin/bash
# Run on the GPU box (gpurun):  tools/collect_profiles.sh r01
# 1) rocprofv3 --kernel-trace --stats of the default bench command, 2) two PMC passes (FETCH_SIZE; WRITE_SIZE) of a
# short bench run, each in its own process with --kernel-trace only.  Raw CSVs land in gpurun_out/profiles_<tag>/;
# tools/summarize_profiles.py turns them into the files committed under profiles/.
set -o pipefail
tag=${1:-r01}
shift || true
extra="$@"          # further bench.py arguments, e.g.  tools/collect_profiles.sh r02_512 --resolution 512 --batch 8
out=gpurun_out/profiles_$tag
mkdir -p $out
cd /tmp; export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
legs="--no-cpu-baseline --no-fp32-leg --no-h2d-leg"      # the timed region only: the extra legs would add iterations to the trace
python bench.py $legs $extra > $out/bench_plain.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -- python bench.py $legs $extra > $out/bench_stats.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python bench.py --steps 2 --warmup 1 $legs --no-kernel-clock $extra > $out/bench_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/write -- python bench.py --steps 2 --warmup 1 $legs --no-kernel-clock $extra > $out/bench_write.log 2>&1
python tools/summarize_profiles.py $tag --dst $out/summary > $out/summary.log 2>&1
# the raw traces are tens of MB each; gpurun only copies 64 MiB back
find $out -name "*kernel_trace.csv" -delete; find $out -name "*counter_collection.csv" -delete
echo "collected into $out (summaries in $out/summary)"

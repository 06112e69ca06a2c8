#!/bin/bash
# rocprofv3 passes over the DRIVER'S bench command (headline + every secondary config), on the GPU box through gpurun:
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_bench.sh r03'
# One --kernel-trace --stats pass and one pass per counter (FETCH_SIZE and WRITE_SIZE do not fit one pass; counters are never
# combined with trace domains).  Summarised into profiles/ by tools/summarize_bench_prof.py.
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/${TAG}prof
mkdir -p $O
B="python3 bench.py --steps 20 --warmup 5 --cpu-sample 0"
python3 bench.py --steps 20 --warmup 5 > $O/bench_plain.json 2> $O/bench_plain.err || echo "plain bench failed"
echo "plain done"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $B > $O/stats.json 2> $O/stats.err || echo "stats failed"
echo "stats done"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -o run -- $B > $O/fetch.json 2> $O/fetch.err || echo "fetch failed"
echo "fetch done"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -o run -- $B > $O/write.json 2> $O/write.err || echo "write failed"
echo "write done"
find $O -name "*.db" -delete
# the per-dispatch counter CSVs are large (one row per launch): keep what the summary needs
python3 tools/summarize_bench_prof.py $O $TAG --compact || echo "summary failed"
du -sh $O

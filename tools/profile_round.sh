#!/bin/bash
# Round-2 profiling session on the GPU box (run through gpurun): rocprofv3 kernel traces and PMC passes for the bench line,
# WMF C4 and the RelMF tile schedule.  Counters are collected in their own passes (FETCH_SIZE and WRITE_SIZE do not fit one).
#   /usr/local/graft/bin/gpurun --timeout 1100 -- 'bash tools/profile_round.sh'
set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r02prof
mkdir -p $O
B="python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-secondary"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/bench_stats -o run -- $B > $O/bench_stats.log 2>&1 || echo "bench stats failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/bench_fetch -o run -- $B > $O/bench_fetch.log 2>&1 || echo "bench fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/bench_write -o run -- $B > $O/bench_write.log 2>&1 || echo "bench write failed"
echo "bench passes done"
W="python3 tools/bench_models.py wmf"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/wmf_stats -o run -- $W > $O/wmf_stats.log 2>&1 || echo "wmf stats failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/wmf_fetch -o run -- $W > $O/wmf_fetch.log 2>&1 || echo "wmf fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/wmf_write -o run -- $W > $O/wmf_write.log 2>&1 || echo "wmf write failed"
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O/wmf_mfma -o run -- $W > $O/wmf_mfma.log 2>&1 || echo "wmf mfma failed"
echo "wmf passes done"
R="python3 tools/relmf_check.py speed"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/relmf_stats -o run -- $R > $O/relmf_stats.log 2>&1 || echo "relmf stats failed"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/relmf_fetch -o run -- $R > $O/relmf_fetch.log 2>&1 || echo "relmf fetch failed"
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/relmf_write -o run -- $R > $O/relmf_write.log 2>&1 || echo "relmf write failed"
echo "relmf passes done"
# keep what travels back small: the per-dispatch CSVs are enough
find $O -name "*.db" -delete
du -sh $O

#!/bin/bash
# usage: tools/rocprof_stats.sh <tag> <bench args...>
# Runs bench.py under rocprofv3 --kernel-trace --stats on the GPU box and keeps only the small
# kernel-stats CSV (gpurun_out/<tag>_kernel_stats.csv) plus the bench line.
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/prof_$TAG
timeout -k 5 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_$TAG -o $TAG -- python "$REPO/bench.py" "$@" > "$REPO/gpurun_out/${TAG}_bench.log" 2>&1
find /tmp/prof_$TAG -name "*kernel_stats.csv" -exec cp {} "$REPO/gpurun_out/${TAG}_kernel_stats.csv" \;
grep '^{' "$REPO/gpurun_out/${TAG}_bench.log" | tail -1
head -40 "$REPO/gpurun_out/${TAG}_kernel_stats.csv" | cut -c1-200

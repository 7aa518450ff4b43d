#!/bin/bash
# usage: tools/rocprof_pmc.sh <tag> "<COUNTER ...>" <command...>
# One rocprofv3 --pmc pass (counters must fit one pass; run again for others). Keeps a per-kernel
# average of every counter in gpurun_out/<tag>_pmc.csv.
set -u
TAG=$1; CTR=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
timeout -k 5 240 rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d /tmp/pmc_$TAG -o $TAG -- "$@" > "$REPO/gpurun_out/${TAG}_pmc.log" 2>&1
F=$(find /tmp/pmc_$TAG -name "*counter_collection.csv" | head -1)
python3 - "$F" "$REPO/gpurun_out/${TAG}_pmc.csv" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        k = (r["Kernel_Name"][:80], r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
with open(sys.argv[2], "w") as o:
    o.write("kernel,counter,launches,avg_value,total\n")
    for (k, c), (s, n) in sorted(acc.items()):
        o.write(f'"{k}",{c},{n},{s/n:.1f},{s:.1f}\n')
print(open(sys.argv[2]).read()[:6000])
PY

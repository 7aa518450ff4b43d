#!/bin/bash
# usage: tools/rocprof_pmc.sh <tag> "<COUNTER ...>" <command...>
# One rocprofv3 --pmc pass (counters must fit one pass; run again for others).  Keeps a per-kernel average of every
# counter in gpurun_out/<tag>_pmc.csv; every row also carries the kernel's average duration (ns, from the kernel
# trace of the SAME pass) and, when GRBM_GUI_ACTIVE is among the counters, the derived clock
# (GRBM_GUI_ACTIVE / 8 XCDs / duration) -- so MFMA-busy fractions and TFLOP/s of one row refer to one launch.
set -u
TAG=$1; CTR=$2; shift 2
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_$TAG
timeout -k 5 ${PMC_TIMEOUT:-240} rocprofv3 --kernel-trace --pmc $CTR --output-format csv -d /tmp/pmc_$TAG -o $TAG -- "$@" > "$REPO/gpurun_out/${TAG}_pmc.log" 2>&1
F=$(find /tmp/pmc_$TAG -name "*counter_collection.csv" | head -1)
K=$(find /tmp/pmc_$TAG -name "*kernel_trace.csv" | head -1)
python3 - "$F" "$REPO/gpurun_out/${TAG}_pmc.csv" "${K:-}" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
dur = collections.defaultdict(lambda: [0.0, 0])
names = {}
durs_by_id = {}
if len(sys.argv) > 3 and sys.argv[3]:
    with open(sys.argv[3]) as f:
        for r in csv.DictReader(f):
            try:
                durs_by_id[r["Dispatch_Id"]] = (float(r["End_Timestamp"]) - float(r["Start_Timestamp"]), r["Kernel_Name"])
            except (KeyError, ValueError):
                pass
seen = set()
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        name = r["Kernel_Name"][:100]
        k = (name, r["Counter_Name"])
        acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
        did = r.get("Dispatch_Id")
        key = (name, did)
        if key in seen:
            continue
        seen.add(key)
        d = None
        if "Start_Timestamp" in r and "End_Timestamp" in r and r["Start_Timestamp"] and r["End_Timestamp"]:
            d = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
        elif did in durs_by_id:
            d = durs_by_id[did][0]
        if d is not None and d > 0:
            dur[name][0] += d; dur[name][1] += 1
with open(sys.argv[2], "w") as o:
    o.write("kernel,counter,launches,avg_value,total,avg_duration_ns,derived_clock_GHz\n")
    for (k, c), (s, n) in sorted(acc.items()):
        d = dur[k][0] / dur[k][1] if dur[k][1] else 0.0
        clk = ""
        if c == "GRBM_GUI_ACTIVE" and d > 0:
            clk = f"{(s / n) / 8.0 / d:.3f}"
        o.write(f'"{k}",{c},{n},{s/n:.1f},{s:.1f},{d:.0f},{clk}\n')
print(open(sys.argv[2]).read()[:8000])
PY

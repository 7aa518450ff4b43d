#!/bin/bash
# usage: tools/trace_step.sh <tag> [bench args...]
# rocprofv3 --kernel-trace of a short bench run; keeps the ORDERED launch sequence of the last timed step as a compact
# text file gpurun_out/<tag>_step_trace.txt (start offset us, duration us, queue, kernel name) plus per-queue busy time
# and the gaps between consecutive launches of the main queue: the input of the launch-diet work (which small launches
# exist, where the stream idles).
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
ARGS="${*:---steps 3 --warmup 3 --no-cpu-baseline --no-kernel-timing --no-arith-ab --also none}"
rm -rf /tmp/trs_$TAG
timeout -k 5 400 rocprofv3 --kernel-trace --output-format csv -d /tmp/trs_$TAG -o $TAG -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/${TAG}_trace_bench.log" 2>&1
python3 - "$TAG" "$REPO/gpurun_out/${TAG}_step_trace.txt" <<'PY'
import csv, glob, sys, collections
tag, out = sys.argv[1:3]
rows = []
for kf in glob.glob(f"/tmp/trs_{tag}/**/*kernel_trace.csv", recursive=True):
    with open(kf) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"]))
rows.sort()
# steps are delimited by the optimizer kernel (one k_adam / k_rmsprop per step): keep the launches between the last two
marks = [i for i, r in enumerate(rows) if "k_adam" in r[3] or "k_rmsprop" in r[3]]
if len(marks) >= 2:
    rows = rows[marks[-2] + 1: marks[-1] + 1]
t0 = rows[0][0]
busy = collections.defaultdict(float)
cnt = collections.Counter()
with open(out, "w") as o:
    o.write(f"# {len(rows)} launches, span {(rows[-1][1] - t0) / 1e3:.1f} us\n")
    prev_end = {}
    gaps = collections.defaultdict(float)
    for s, e, q, n in rows:
        gap = (s - prev_end[q]) / 1e3 if q in prev_end else 0.0
        if gap > 0:
            gaps[q] += gap
        prev_end[q] = max(prev_end.get(q, 0), e)
        busy[q] += (e - s) / 1e3
        short = n.split("(")[0][-70:] if n.startswith("void") or "::" in n else n[:70]
        cnt[short] += 1
        o.write(f"{(s - t0) / 1e3:10.1f} {(e - s) / 1e3:8.1f} gap{gap:7.1f} q{q} {short}\n")
    o.write("# busy us per queue: " + ", ".join(f"q{q}={v:.0f}" for q, v in busy.items()) + "\n")
    o.write("# idle gaps us per queue: " + ", ".join(f"q{q}={v:.0f}" for q, v in gaps.items()) + "\n")
    o.write("# launches per kernel:\n")
    for k, v in cnt.most_common():
        o.write(f"#   {v:4d} {k}\n")
print(open(out).read()[-3000:])
PY

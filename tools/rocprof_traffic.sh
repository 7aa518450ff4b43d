#!/bin/bash
# usage: tools/rocprof_traffic.sh <tag> [bench args...]
# HBM traffic per kernel of `python bench.py <args>` as /opt/skills/guides/MI355X_MICROARCH.md (HBM section)
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE rocprofv3 --pmc passes (they do not fit one pass), no trace
# domain other than --kernel-trace, FETCH_SIZE doubled (gfx950 tallies 128-B requests of wide coalesced reads at 64 B),
# sizes in KiB.  Writes gpurun_out/<tag>_traffic.json:
#   {"source": ..., "kernels": {name: {launches, fetch_bytes_avg (x2 applied), write_bytes_avg, hbm_bytes_per_launch,
#                                      total_hbm_bytes, avg_duration_ns}}}
# Copy it to profiles/<round>_traffic.json and profiles/traffic_latest.json (bench.py reads the latter for
# `roofline.traffic`).
set -u
TAG=$1; shift
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
mkdir -p "$REPO/gpurun_out"
cd /tmp && export TMPDIR=/tmp
ARGS="${*:---steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing --no-arith-ab --also none}"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/tr_${TAG}_$C
  timeout -k 5 ${PMC_TIMEOUT:-400} rocprofv3 --kernel-trace --pmc $C --output-format csv -d /tmp/tr_${TAG}_$C -o $TAG -- python "$REPO/bench.py" $ARGS > "$REPO/gpurun_out/${TAG}_traffic_$C.log" 2>&1
done
CODE_ID=$(cd "$REPO" && python3 -m pose2mesh_release_amd.build --source-id 2>/dev/null | tail -1)
python3 - "$TAG" "$REPO/gpurun_out/${TAG}_traffic.json" "$ARGS" "$CODE_ID" <<'PY'
import csv, glob, json, sys, collections
tag, out, args, code_id = sys.argv[1:5]
rec = collections.defaultdict(lambda: {"FETCH_SIZE": [0.0, 0], "WRITE_SIZE": [0.0, 0], "dur": [0.0, 0]})
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(f"/tmp/tr_{tag}_{c}/**/*counter_collection.csv", recursive=True)
    durs = {}
    for kf in glob.glob(f"/tmp/tr_{tag}_{c}/**/*kernel_trace.csv", recursive=True):
        with open(kf) as f:
            for r in csv.DictReader(f):
                try:
                    durs[r["Dispatch_Id"]] = float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
                except (KeyError, ValueError):
                    pass
    for fn in files:
        with open(fn) as f:
            for r in csv.DictReader(f):
                if r["Counter_Name"] != c:
                    continue
                k = r["Kernel_Name"]
                rec[k][c][0] += float(r["Counter_Value"]); rec[k][c][1] += 1
                d = durs.get(r.get("Dispatch_Id"))
                if d:
                    rec[k]["dur"][0] += d; rec[k]["dur"][1] += 1
kern = {}
for k, v in rec.items():
    nf, nw = v["FETCH_SIZE"][1], v["WRITE_SIZE"][1]
    if nf == 0 and nw == 0:
        continue
    fetch = 2.0 * 1024.0 * v["FETCH_SIZE"][0] / max(nf, 1)      # KiB -> bytes, x2 gfx950 correction
    write = 1024.0 * v["WRITE_SIZE"][0] / max(nw, 1)
    n = max(nf, nw)
    kern[k] = {"launches": n, "fetch_bytes_avg": fetch, "write_bytes_avg": write, "hbm_bytes_per_launch": fetch + write,
               "total_hbm_bytes": (fetch + write) * n,
               "avg_duration_ns": v["dur"][0] / v["dur"][1] if v["dur"][1] else None}
json.dump({"code_id": code_id, "source": f"rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE (two passes) -- python bench.py {args}",
           "note": "fetch_bytes_avg already carries the x2 gfx950 correction of MI355X_MICROARCH.md (HBM section)",
           "kernels": kern}, open(out, "w"), indent=1, sort_keys=True)
top = sorted(kern.items(), key=lambda kv: -kv[1]["total_hbm_bytes"])[:25]
for k, v in top:
    d = v["avg_duration_ns"] or 0
    print(f"{v['launches']:5d} x {v['hbm_bytes_per_launch']/1e6:10.1f} MB/launch  {d/1e3:9.1f} us  "
          f"{(v['hbm_bytes_per_launch']/d if d else 0):7.2f} GB/s(x1e0=B/ns)  {k[:110]}")
PY

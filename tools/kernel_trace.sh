#!/bin/bash
# Per-launch durations of the march at several sample counts: one rocprofv3 kernel trace per count, the k_sdf_march
# launches of the last traced frame printed in order (ms).  usage: tools/kernel_trace.sh [spp ...]
cd /tmp && export TMPDIR=/tmp
for spp in ${@:-64 128 256 1024}; do
rm -rf /tmp/kt_$spp
timeout -s KILL 300 rocprofv3 --kernel-trace -d /tmp/kt_$spp -o t --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline --spp $spp > $GRAFT_REPO_ROOT/gpurun_out/kt_$spp.log 2>&1 || exit 1
python3 - $spp $(find /tmp/kt_$spp -name "*kernel_trace.csv" | head -1) <<'PY'
import csv, sys
spp, path = sys.argv[1], sys.argv[2]
rows = list(csv.DictReader(open(path)))
name = [k for k in rows[0] if k.lower() == "kernel_name"][0]
s = [k for k in rows[0] if k.lower().startswith("start")][0]
e = [k for k in rows[0] if k.lower().startswith("end")][0]
for kern in ("k_sdf_march", "k_sort_keys", "k_sort_scan", "k_sort_scatter", "k_shade_sdf", "k_extend", "k_shade(", "k_generate"):
    d = [(int(r[s]), (int(r[e]) - int(r[s])) / 1e6) for r in rows if kern in r[name]]
    d.sort()
    per_frame = len(d) // 3   # three frames were traced (1 warm-up + 2)
    last = [round(x[1], 2) for x in d[-per_frame:]]
    print(f"spp {spp} {kern:12s} sum {sum(last):7.2f} ms  launches {last}")
allk = sorted((int(r[s]), int(r[e])) for r in rows)
PY
done

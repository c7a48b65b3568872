#!/bin/bash
# thin-history cases: hypothesis counts per level and the kernel timeline of one decode
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r4px10; mkdir -p $O
for c in "8320x2048 q90" "8320x2048 q75"; do
  tag=$(echo "$c" | tr ' ' '_')
  MIJ_PX_DEBUG=1 MIJ_PX_COUNTS=1 timeout -k 10 300 python3 tools/r4_px_test.py "$c" 2> $O/counts_$tag.txt | grep synth
done
cd /tmp && export TMPDIR=/tmp
for c in "8320x2048 q90" "8320x2048 q75"; do
  tag=$(echo "$c" | tr ' ' '_')
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $O/prof_$tag -o px --output-format csv -- python3 $GRAFT_REPO_ROOT/tools/r4_px_test.py "$c" > $O/prof_$tag.log 2>&1
  python3 - "$O/prof_$tag" > $O/timeline_$tag.txt <<'PY'
import csv, glob, sys
t = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
rows = list(csv.DictReader(open(t[0])))
rows = [r for r in rows if "k_px" in r["Kernel_Name"] or "k_scan_decode_wave" in r["Kernel_Name"] or "k_unstuff" in r["Kernel_Name"] or "k_idct" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the last decode = after the last k_idct of the first
idct = [i for i, r in enumerate(rows) if "k_idct" in r["Kernel_Name"]]
half = rows[idct[-2] + 1:] if len(idct) >= 2 else rows
t0 = int(half[0]["Start_Timestamp"])
byq = {}
for r in half:
    byq.setdefault(r["Queue_Id"], []).append(r)
for q, rs in byq.items():
    print("queue %s: %.2f ms busy, %d kernels, ends at %.2f ms" % (q, sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rs) / 1e6, len(rs), (max(int(r["End_Timestamp"]) for r in rs) - t0) / 1e6))
for q, rs in byq.items():
    print("== queue", q)
    for r in rs:
        d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
        if d >= 0.5:
            print("  %8.2f ms at %8.2f ms %s grid %s" % (d, (int(r["Start_Timestamp"]) - t0) / 1e6, r["Kernel_Name"].split("(")[0][-34:], r.get("Grid_Size_X", "?")))
PY
  rm -rf $O/prof_$tag
done

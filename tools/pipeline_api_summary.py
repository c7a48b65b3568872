"""Steady-state runtime calls per image from the HIP-API trace of tools/profile_pipeline_api.sh (second half of the kernel launches)."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ls = [r for r in rows if r["Function"] == "hipLaunchKernel"]
a, b = int(ls[len(ls) // 2]["Start_Timestamp"]), int(ls[-50]["Start_Timestamp"])
win = [r for r in rows if a <= int(r["Start_Timestamp"]) <= b]
nl = sum(1 for r in win if r["Function"] == "hipLaunchKernel")
images = nl / 6.0     # K1, K3, K4, K5 (two kernels), K6
print("steady-state window: %.1f ms, %d kernel launches = %.0f images (six launches per image)" % ((b - a) / 1e6, nl, images))
c = collections.Counter(r["Function"] for r in win)
for k in ("hipStreamSynchronize", "hipDeviceSynchronize", "hipEventSynchronize", "hipMemcpy", "hipMemcpyDtoH", "hipMemcpyWithStream", "hipStreamWaitEvent"):
    c.setdefault(k, 0)
for k, v in sorted(c.items(), key=lambda kv: -kv[1]):
    print("%-40s %6d  %.2f per image" % (k, v, v / images))

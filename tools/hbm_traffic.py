"""HBM traffic per kernel launch from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE cannot share a pass).

    tools/pmc_pass.sh fetch FETCH_SIZE && tools/pmc_pass.sh write WRITE_SIZE        # on the GPU box
    python tools/hbm_traffic.py gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/rNN_hbm_traffic.json

Corrections (MI355X_MICROARCH.md, HBM section): both counters are reported in KiB; on gfx950 FETCH_SIZE tallies the
128-B requests of wide streaming reads at 64 B, so it is doubled; WRITE_SIZE is exact for 16-B-per-lane stores.
bench.py copies `total_bytes` of the dominant kernel into roofline.traffic when the workload matches AND the
`library_source_hash` recorded here (read from the bench line each pass logged, i.e. the hash compiled into the libmijpeg.so
that ran under the counters) equals the hash of the library bench.py itself has loaded: counters taken on other kernels are
never quoted.
"""
import collections
import csv
import glob
import json
import re
import sys

SHORT = {"k_transform": "k_transform", "k_encode": "k_encode", "k_compact": "k_compact", "k_dc_stats": "k_dc_stats",
         "k_build_tables": "k_build_tables", "k_scan_chunks": "k_scan_chunks", "k_scan_totals": "k_scan_totals"}


NOSTATS = re.compile(r"k_transform<[^>]*,\s*(false|0)\s*>")


def per_kernel(directory, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(directory + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] != counter:
                continue
            name = row["Kernel_Name"]
            for key in SHORT:
                if key in name:
                    # the transform without the fused statistics (last template argument false): stage A alone, its own entry
                    if key == "k_transform" and NOSTATS.search(name):
                        key = "k_transform_nostats"
                    acc[key].append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


def logged_hash(directory):
    """library_source_hash of the bench line the pass wrote to <directory>/log.txt (tools/pmc_pass.sh)."""
    try:
        lines = [l for l in open(directory + "/log.txt") if l.startswith("{")]
        return json.loads(lines[-1]).get("library_source_hash")
    except (OSError, ValueError, IndexError):
        return None


def main():
    fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
    write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
    h_f, h_w = logged_hash(sys.argv[1]), logged_hash(sys.argv[2])
    if h_f is None or h_f != h_w:
        sys.exit("the two passes do not name one library_source_hash (%s / %s): not writing a traffic file" % (h_f, h_w))
    # optional third pass (tools/pmc_pass.sh b ... SQ_INSTS_VALU ...): vector wave-instructions per launch
    valu = {}
    if len(sys.argv) > 3 and logged_hash(sys.argv[3]) == h_f:
        valu, _ = per_kernel(sys.argv[3], "SQ_INSTS_VALU")
    # optional fourth pass (tools/pmc_pass.sh a SQ_WAVES ...): waves per launch (tools/valu_bound.py solves a kernel's loop trip count from both)
    nwaves = {}
    if len(sys.argv) > 4 and logged_hash(sys.argv[4]) == h_f:
        nwaves, _ = per_kernel(sys.argv[4], "SQ_WAVES")
    kernels = {}
    for k in sorted(set(fetch) | set(write)):
        fb = int(fetch.get(k, 0.0) * 1024 * 2)
        wb = int(write.get(k, 0.0) * 1024)
        kernels[k] = {"fetch_bytes": fb, "write_bytes": wb, "total_bytes": fb + wb, "launches_averaged": [nf.get(k, 0), nw.get(k, 0)]}
        if k in valu:
            kernels[k]["valu_wave_instructions"] = int(valu[k])
        if k in nwaves:
            kernels[k]["waves"] = int(nwaves[k])
    print(json.dumps({
        "library_source_hash": h_f,
        "workload": "8320x40000 q95 4:2:2 optimised, AUTO restart interval (64 MCUs), 1 GPU (bench.py defaults)",
        "method": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes over `bench.py --steps 2 --warmup 1 --kernel-pass 2 --stage-a-pass 2`; "
                  "KiB -> bytes; FETCH_SIZE x2 (gfx950 counts 128-B read requests as 64 B); mean per launch. k_transform_nostats = "
                  "k_transform<..., false>, the transform without the fused statistics (stage A alone; the bench's stage-A pass launches it)",
        "kernels": kernels}, indent=1))


if __name__ == "__main__":
    main()

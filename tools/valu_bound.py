#!/usr/bin/env python3
"""Opcode-class-weighted vector-issue BOUND for the headline kernels (round 5, what replaces the flat "4 cycles per vector instruction"
floor of rounds 3-4, which K4 already ran under).

What the hardware does (profiles/r05_valu_rate.txt = tools/valu_rate.hip on an MI355X, 8 waves per SIMD, independent chains): a wave64 vector
instruction occupies its SIMD-32 for 2 cycles when it is one of  v_add/sub(rev)_u32, v_and/or/xor_b32, v_ashrrev_i32, v_mov_b32,
v_add/mul/fma_f32  and another wave is there to take the next slot; for 4 cycles otherwise (shifts, v_lshl_add / v_add3 / v_or3, every v_cvt,
v_cmp, v_perm, v_alignbit, v_bfe / v_bfi, 24-bit and 32-bit multiplies, v_min / v_max / v_med3, v_dot2 / v_dot4, everything packed, SDWA forms,
64-bit shifts); v_mad_i16 takes 8. The SQ counters do not tell the classes apart (SQ_ACTIVE_INST_VALU and SQ_THREAD_CYCLES_VALU / 64 count one
unit per instruction whatever its class: profiles/r05_valu_rate_pmc.txt), so the class MIX comes from the kernel's ISA and the COUNT from
SQ_INSTS_VALU:

    bound_ms = SQ_INSTS_VALU x (2 x f2 + 4 x f4 + 8 x f8) / (1,024 SIMDs x shader clock)

f2 / f4 / f8 = the shares of the classes among the vector instructions of the kernel's basic blocks, each block weighted by how often it runs:
blocks outside any loop once per wave, blocks inside a loop T times, T being the one trip count that reproduces the measured SQ_INSTS_VALU
(`--valu kernel=count --waves kernel=count`; without them T = 1000, i.e. the mix of the loop bodies). The big straight-line blocks (a tile's
fast path in K1, a chunk of eight coefficients in K4) carry nearly all the weight either way; rare paths inside the loops (edge tiles, ZRL
re-coding) have the same kind of mix, so the shares move by about a point with the choice. An instruction that was never measured counts as 2 cycles
(v_cndmask_b32 among them: the tool's chain through vcc measured the dependency, not the issue cost): the bound errs low, as a bound must.

Usage:  python tools/valu_bound.py [--asm FILE.s] [--valu NAME=N ...] [--waves NAME=N ...] > profiles/rNN_valu_classes.json
Without --asm the device assembly is produced with the library's own compile flags (hipcc -S, ~40 s)."""
import argparse
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

# kernels of the headline step: short name -> regular expression on the demangled name in the "; -- Begin function" comment / mangled label
KERNELS = {
    "k_transform": r"k_transformILi2ELi1ELb1ELb1E",
    "k_transform_nostats": r"k_transformILi2ELi1ELb1ELb0E",
    "k_encode": r"k_encodeILi1ELb0ELi16E",
    "k_compact": r"9k_compact",
}
TWO = ("v_add_u32", "v_sub_u32", "v_subrev_u32", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32", "v_addc_co_u32", "v_subb_co_u32", "v_and_b32", "v_or_b32",
       "v_xor_b32", "v_not_b32", "v_ashrrev_i32", "v_mov_b32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_fma_f32", "v_fmac_f32", "v_mac_f32",
       "v_cndmask_b32", "v_accvgpr", "v_nop", "v_readfirstlane_b32", "v_readlane_b32", "v_writelane_b32", "v_mov_b64")
EIGHT = ("v_mad_i16", "v_mad_u16", "v_exp_f32", "v_log_f32", "v_rcp_f32", "v_rsq_f32", "v_sqrt_f32", "v_sin_f32", "v_cos_f32")


def cls(op):
    base = re.sub(r"_(e32|e64|sdwa|dpp)$", "", op)
    if op.endswith("_sdwa") or op.endswith("_dpp"):
        return 4
    if base in EIGHT:
        return 8
    if base in TWO:
        return 2
    return 4


def functions(path):
    lines = open(path).read().split("\n")
    out, i = {}, 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("_Z") and l.rstrip().split(":")[0] and ":" in l:
            name = l.split(":")[0]
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            out[name] = lines[i + 1:j]
            i = j
        i += 1
    return out


def blocks(body):
    res, cur, depth = [], [], 0
    res.append(("entry", 0, cur))
    for l in body:
        if l.startswith(".LBB"):
            m = re.search(r"Depth=(\d+)", l)
            depth = int(m.group(1)) if m else 0
            cur = []
            res.append((l.split(":")[0], depth, cur))
        elif l.startswith("\t") and not l.strip().startswith((".", ";")):
            cur.append(l.strip().split()[0])
    return res


def analyse(body, valu_dyn=None, waves=None):
    bl = blocks(body)
    per = []
    for name, depth, ins in bl:
        c = collections.Counter()
        ops = collections.Counter()
        for op in ins:
            if op.startswith("v_") and not op.startswith(("v_mfma", "v_smfma")):
                c[cls(op)] += 1
                ops[re.sub(r"_(e32|e64)$", "", op)] += 1
        per.append((name, depth, c, ops, len(ins)))
    once = sum(sum(c.values()) for _, d, c, _, _ in per if d == 0)
    loop = sum(sum(c.values()) for _, d, c, _, _ in per if d > 0)
    T = 1000.0
    if valu_dyn and waves and loop:
        T = max(1.0, (valu_dyn / waves - once) / loop)
    tot = collections.Counter()
    opw = collections.Counter()
    for _, d, c, ops, _ in per:
        w = T if d > 0 else 1.0
        for k, n in c.items():
            tot[k] += n * w
        for k, n in ops.items():
            opw[k] += n * w
    s = sum(tot.values()) or 1.0
    f = {k: tot.get(k, 0.0) / s for k in (2, 4, 8)}
    big = sorted(per, key=lambda x: -sum(x[2].values()))[:6]
    return {"static_vector_instructions": once + loop, "in_loops": loop, "loop_trip_count_T": round(T, 2),
            "T_source": "solved from SQ_INSTS_VALU / waves" if (valu_dyn and waves) else "nominal (mix of the loop bodies)",
            "share_2_cycle": round(f[2], 4), "share_4_cycle": round(f[4], 4), "share_8_cycle": round(f[8], 4),
            "cycles_per_vector_instruction_bound": round(2 * f[2] + 4 * f[4] + 8 * f[8], 4),
            "top_opcodes_weighted_share": {k: round(v / s, 4) for k, v in opw.most_common(14)},
            "largest_blocks": [{"block": n, "loop_depth": d, "vector": sum(c.values()), "four_cycle": c.get(4, 0), "instructions": tot_n}
                               for n, d, c, _, tot_n in big]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    ap.add_argument("--valu", action="append", default=[])
    ap.add_argument("--waves", action="append", default=[])
    ap.add_argument("--traffic", help="profiles/rNN_hbm_traffic.json: take SQ_INSTS_VALU per launch from it (same library hash only)")
    args = ap.parse_args()
    from nvjpeg_imagecompressor_amd import build as B
    asm = args.asm
    if not asm:
        asm = os.path.join(tempfile.gettempdir(), "mij_kernels_%s.s" % B.source_hash()[:12])
        if not os.path.exists(asm):
            subprocess.check_call([B._hipcc()] + B.FLAGS + ["--cuda-device-only", "-S", os.path.join(B.CSRC, "mij_kernels.hip"), "-o", asm],
                                  stderr=subprocess.DEVNULL)
    valu = dict((a.split("=")[0], float(a.split("=")[1])) for a in args.valu)
    waves = dict((a.split("=")[0], float(a.split("=")[1])) for a in args.waves)
    if args.traffic:
        t = json.load(open(args.traffic))
        if t.get("library_source_hash") == B.source_hash():
            for k, v in t.get("kernels", {}).items():
                if "valu_wave_instructions" in v:
                    valu.setdefault(k, float(v["valu_wave_instructions"]))
                if "waves" in v:
                    waves.setdefault(k, float(v["waves"]))
    fn = functions(asm)
    out = {"library_source_hash": B.source_hash(), "method": __doc__.split("\n\n")[1].replace("\n", " "),
           "rate_table": "profiles/r05_valu_rate.txt", "kernels": {}}
    for short, pat in KERNELS.items():
        names = [n for n in fn if re.search(pat, n)]
        if not names:
            continue
        out["kernels"][short] = dict(analyse(fn[names[0]], valu.get(short), waves.get(short)), symbol=names[0])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()

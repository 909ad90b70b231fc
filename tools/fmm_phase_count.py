#!/usr/bin/env python
"""Static instruction counts of the eikonal kernel's marching loop, phase by phase.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S -DDZ_FMM_MARK -Iinclude \
          -o /tmp/fmm_mark.s dazimsurftomo_amd/csrc/fmm.hip
    python tools/fmm_phase_count.py /tmp/fmm_mark.s [kernel-substring] [--blocks]

The DZ_FMM_MARK build puts `; MARK n` comments where the DZ_FMM_PROF build reads the clock (march() in fmm.hip): 7 = loop top,
0 = after the stencil loads are issued, 1 = after the sift-down, 2 = (wait), 3 = after the loads are consumed and the slot
look-up reads are issued, 4 = after the quadrant solve and the look-up, 5 = after the owner-lane updates (fast path), 6 = after
the sequential / parallel rise rounds (slow path).  march() is inlined once per call site; every instance is reported.

Per region: instructions by issue class as measured by tools/valu_issue_calib.hip (profiles/r5_valu_issue.md):
  v2 = VALU, 2 cycles per wave64 instruction with >= 2 wavefronts on the SIMD (v_add/sub/mul_f32, v_add/sub_u32, v_and/or/xor_b32, v_mov_b32)
  v4 = VALU, 4 cycles (compares, v_cndmask, DPP, shifts, min/max, three-register VOP3, 64-bit and packed operations, v_readlane)
  v8 = transcendental (v_rcp_f32, v_sqrt_f32, v_rsq_f32)
  s  = scalar ALU / branches / waits,  lds = ds_*,  vm = global / buffer / flat memory
A region's blocks are listed in layout order; blocks that the usual pop does not execute (wave-uniform rare paths) are marked by
hand in profiles/r5_fmm_phase_split.md, which is made from this output."""
import re
import sys
from collections import Counter, OrderedDict

V2 = {"v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32", "v_add_u32", "v_sub_u32", "v_subrev_u32", "v_and_b32", "v_or_b32", "v_xor_b32",
      "v_mov_b32", "v_add_co_u32", "v_sub_co_u32", "v_subrev_co_u32"}
V8 = {"v_rcp_f32", "v_sqrt_f32", "v_rsq_f32", "v_rcp_iflag_f32", "v_exp_f32", "v_log_f32", "v_sin_f32", "v_cos_f32"}


def klass(op, text):
    if op.startswith("v_"):
        base = op
        for suf in ("_e32", "_e64", "_dpp", "_sdwa"):
            if base.endswith(suf):
                base = base[: -len(suf)]
        if base in V8:
            return "v8"
        if "dpp" in op or "quad_perm" in text or "row_" in text or "sdwa" in op:
            return "v4"
        if base in V2:
            return "v2"
        if base == "v_fma_f32" and re.search(r",\s*-?\d+(\.\d+)?\s*$", text):   # inline constant as third source: measured 2 cycles
            return "v2"
        return "v4"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vm"
    if op.startswith("s_"):
        return "s"
    return "other"


def main(path, want, show_blocks):
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S*fmm_kernel\S*:", l) and want in l:
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found: " + want)
    end = next(i for i in range(start, len(lines)) if "s_endpgm" in lines[i] and i > start + 100)
    # walk: regions between MARKs; each region = list of (block label, Counter)
    regions = []   # (from_mark, to_mark, [(label, Counter, ops)])
    cur_mark = None
    blocks = [("entry", Counter(), [])]
    for l in lines[start + 1 : end + 1]:
        t = l.strip()
        m = re.match(r"; MARK (\d+)", t)
        if m:
            regions.append((cur_mark, int(m.group(1)), blocks))
            cur_mark = int(m.group(1))
            blocks = [("(cont)", Counter(), [])]
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", t)
        if m:
            blocks.append((m.group(1), Counter(), []))
            continue
        if not t or t.startswith((";", ".", "//")):
            continue
        op = t.split()[0]
        k = klass(op, t.split(";")[0])
        blocks[-1][1][k] += 1
        blocks[-1][2].append(op)
    regions.append((cur_mark, None, blocks))
    names = {(7, 0): "root, coordinates, addresses, stencil loads issued", (0, 1): "sift-down (pop_root_par)", (1, 2): "(wait marker)",
             (2, 3): "loads consumed, dropped entry's word, status, slot look-up reads issued", (3, 4): "quadrant solve + slot look-up",
             (4, 5): "owner-lane updates (fast path, one-level rise)", (5, 6): "rise rounds (slow path)", (6, 7): "loop back"}
    inst = 0
    for a, b, blks in regions:
        if a == 7 and b == 0:
            inst += 1
            print(f"\n## march() instance {inst}\n")
            print("| phase | v2 | v4 | v8 | s | lds | vm | VALU total | issue cycles (2/4/8) |")
            print("|---|---:|---:|---:|---:|---:|---:|---:|---:|")
        if a is None or b is None or inst == 0:
            continue
        tot = Counter()
        for _, c, _ in blks:
            tot.update(c)
        valu = tot["v2"] + tot["v4"] + tot["v8"]
        print(f"| {a}->{b} {names.get((a, b), '')} | {tot['v2']} | {tot['v4']} | {tot['v8']} | {tot['s']} | {tot['lds']} | {tot['vm']} | {valu} | "
              f"{2 * tot['v2'] + 4 * tot['v4'] + 8 * tot['v8']} |")
        if show_blocks:
            for lab, c, ops in blks:
                if sum(c.values()):
                    print(f"|   `{lab}` | {c['v2']} | {c['v4']} | {c['v8']} | {c['s']} | {c['lds']} | {c['vm']} | {c['v2'] + c['v4'] + c['v8']} | "
                          f"{' '.join(o for o in ops if o.startswith(('s_cbranch', 's_branch')))} |")


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    main(args[0], args[1] if len(args) > 1 else "fmm_kernelILi512ELb0EtLb1ELi16E", "--blocks" in sys.argv)

#!/usr/bin/env python3
"""Instruction mix of a gfx950 kernel from hipcc's -save-temps assembly.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -save-temps -c csrc/conv_brick16.hip
    python tools/isa_mix.py conv_brick16-hip-amdgcn-amd-amdhsa-gfx950.s 'brick16_conv_kernelILi64ELi0ELi0ELi4E'

For the named kernel: register / spill summary, then per natural loop (a backward s_cbranch to a label; innermost first) the counts of
v_mfma, LDS reads / writes, global / buffer accesses, the other VALU instructions BY OPCODE, SALU, s_waitcnt, s_barrier, and the
non-MFMA VALU per MFMA ratio the stall table of round 4 reported from counters (SQ_INSTS_VALU / SQ_INSTS_MFMA).  Straight-line code
outside loops (prologue, epilogue) is reported the same way.  Static counts: a conditional region inside a loop is counted once.
"""
import collections
import re
import sys


def classify(op):
    if op.startswith("v_mfma") or op.startswith("v_smfmac"):
        return "mfma"
    if op.startswith("ds_read") or op.startswith("ds_load"):
        return "lds_read"
    if op.startswith("ds_write") or op.startswith("ds_store"):
        return "lds_write"
    if op.startswith("ds_"):
        return "lds_other"
    if op.startswith(("global_load_lds", "buffer_load_lds")):
        return "lds_dma"
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        return "vmem_load"
    if op.startswith(("global_store", "buffer_store", "flat_store", "scratch_store")):
        return "vmem_store"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem_other"
    if op.startswith("s_waitcnt"):
        return "s_waitcnt"
    if op.startswith("s_barrier"):
        return "s_barrier"
    if op.startswith("s_nop"):
        return "s_nop"
    if op.startswith(("s_load", "s_buffer_load")):
        return "smem"
    if op.startswith(("s_cbranch", "s_branch")):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "valu_lane"      # SGPR spill traffic / scalarisation
    if op.startswith(("v_accvgpr", )):
        return "valu_acc"
    if op.startswith("v_"):
        return "valu"
    return "other"


def parse_kernel(lines, name_re):
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^_Z\S*:", l) and re.search(name_re, l):
            start = i
            break
    if start is None:
        raise SystemExit(f"kernel matching {name_re!r} not found")
    end = start
    while end < len(lines) and ".end_amdhsa_kernel" not in lines[end] and not lines[end].startswith("\t.section\t.rodata"):
        end += 1
    body = []   # (kind, text): kind 'label' or 'ins'
    for l in lines[start + 1:end]:
        s = l.split(";")[0].rstrip()
        if not s.strip():
            continue
        m = re.match(r"^(\.LBB\S+):", s)
        if m:
            body.append(("label", m.group(1)))
            continue
        t = s.strip()
        if t.startswith(".") or t.endswith(":"):
            continue
        if t.startswith("s_endpgm"):
            body.append(("ins", t))
            break
        body.append(("ins", t))
    return lines[start].split(":")[0], body, (start, end)


def meta(lines, mangled):
    out = {}
    for i, l in enumerate(lines):
        if ".name:" in l and mangled in l and ".kd" not in l:
            for k in range(max(0, i - 70), min(len(lines), i + 12)):
                m = re.match(r"\s*-?\s*\.(vgpr_count|agpr_count|sgpr_count|vgpr_spill_count|sgpr_spill_count|group_segment_fixed_size|private_segment_fixed_size):\s*(\d+)", lines[k])
                if m and abs(k - i) < 12 or (m and m.group(1) == "agpr_count" and k < i and i - k < 70):
                    out[m.group(1)] = int(m.group(2))
            break
    return out


def mix(ins):
    cls = collections.Counter()
    ops = collections.Counter()
    for t in ins:
        op = t.split()[0]
        c = classify(op)
        cls[c] += 1
        if c in ("valu", "valu_lane", "valu_acc", "salu"):
            ops[(c, re.sub(r"_e(32|64)$|_dpp$|_sdwa$", "", op))] += 1
    return cls, ops


def report(title, ins, out):
    cls, ops = mix(ins)
    if not ins:
        return
    nm = cls["mfma"]
    valu = cls["valu"] + cls["valu_lane"] + cls["valu_acc"]
    out.append(f"## {title}: {len(ins)} instructions")
    order = ["mfma", "lds_read", "lds_write", "lds_dma", "vmem_load", "vmem_store", "valu", "valu_lane", "valu_acc", "salu", "smem", "s_waitcnt",
             "s_barrier", "s_nop", "branch", "lds_other", "vmem_other", "other"]
    out.append("   " + "  ".join(f"{k}={cls[k]}" for k in order if cls[k]))
    if nm:
        out.append(f"   non-MFMA VALU per MFMA = {valu / nm:.2f}   SALU per MFMA = {cls['salu'] / nm:.2f}   LDS reads per MFMA = {cls['lds_read'] / nm:.3f}")
    for kind in ("valu", "valu_lane", "valu_acc", "salu"):
        top = [(o, n) for (c, o), n in ops.most_common() if c == kind]
        if top:
            out.append(f"   {kind}: " + ", ".join(f"{o}×{n}" for o, n in top[:14]))


def main():
    path, name_re = sys.argv[1], sys.argv[2]
    lines = open(path).read().splitlines()
    mangled, body, _ = parse_kernel(lines, name_re)
    out = [f"# {mangled}", f"   {meta(lines, mangled)}"]
    labels = {t: i for i, (k, t) in enumerate(body) if k == "label"}
    loops = []
    for i, (k, t) in enumerate(body):
        if k == "ins" and t.startswith(("s_cbranch", "s_branch")):
            tgt = t.split()[-1]
            if tgt in labels and labels[tgt] < i:
                loops.append((labels[tgt], i, tgt))
    loops.sort(key=lambda x: x[1] - x[0])
    inloop = [False] * len(body)
    for lo, hi, tgt in loops:
        ins = [t for (k, t) in body[lo:hi + 1] if k == "ins"]
        inner = sum(1 for a, b, _ in loops if lo < a and b < hi)
        report(f"loop {tgt} (body positions {lo}..{hi}{', contains ' + str(inner) + ' inner loop(s)' if inner else ''})", ins, out)
        for j in range(lo, hi + 1):
            inloop[j] = True
    first = min((lo for lo, _, _ in loops), default=len(body))
    last = max((hi for _, hi, _ in loops), default=-1)
    report("straight-line code before the first loop (prologue)", [t for j, (k, t) in enumerate(body) if k == "ins" and j < first], out)
    report("straight-line code between / outside loops", [t for j, (k, t) in enumerate(body) if k == "ins" and first <= j <= last and not inloop[j]], out)
    report("straight-line code after the last loop (epilogue)", [t for j, (k, t) in enumerate(body) if k == "ins" and j > last], out)
    print("\n".join(out))


if __name__ == "__main__":
    main()

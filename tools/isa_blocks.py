#!/usr/bin/env python3
"""Static instruction census of one kernel in a hipcc -save-temps .s file, per basic block.

usage: isa_blocks.py file.s kernel_symbol_substring [--dump]
Classes: mfma, vpk (packed f32/f16 VALU), vdpp (VALU with a DPP modifier), valu (every other VALU), salu, smem, vmem,
lds, wait (s_waitcnt / s_nop), br (branches).  Used for the per-tile table of profiles/r03_kwtp16_isa_table.md.
"""
import re, sys, collections

def classify(op, line):
    if op.startswith("v_mfma") or op.startswith("v_smfma"): return "mfma"
    if op.startswith("v_"):
        if "dpp" in line or "row_shr" in line or "row_bcast" in line or "wave_shr" in line or "quad_perm" in line: return "vdpp"
        if op.startswith("v_pk_"): return "vpk"
        if op.startswith("v_readlane") or op.startswith("v_readfirstlane") or op.startswith("v_writelane"): return "vlane"
        return "valu"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "wait"
    if op.startswith("s_cbranch") or op.startswith("s_branch") or op.startswith("s_endpgm") or op.startswith("s_setpc"): return "br"
    if op.startswith("s_load") or op.startswith("s_buffer_load") or op.startswith("s_memtime"): return "smem"
    if op.startswith("s_"): return "salu"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_") or op.startswith("scratch_"): return "vmem"
    if op.startswith("ds_"): return "lds"
    return "other"

def main():
    path, sym = sys.argv[1], sys.argv[2]
    dump = "--dump" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if l.startswith("_Z") and sym in l.split(":")[0] and ":" in l:
            start = i; break
    assert start is not None, "kernel not found"
    blocks = collections.OrderedDict()
    cur = "entry"; blocks[cur] = []
    for l in lines[start + 1:]:
        s = l.strip()
        if s.startswith(".Lfunc_end"): break
        if s.startswith(".LBB") and ":" in s.split()[0]:
            cur = s.split(":")[0]; blocks[cur] = []; continue
        if s.startswith("; %bb."):
            cur = s.split(":")[0][2:]; blocks[cur] = []; continue
        if not s or s.startswith(";") or s.startswith("."): continue
        op = s.split()[0]
        blocks[cur].append((op, s))
        if op == "s_endpgm": pass
    tot = collections.Counter()
    print("%-12s %5s | %s" % ("block", "n", "mfma vpk vdpp valu vlane salu smem vmem lds wait br"))
    for b, ins in blocks.items():
        c = collections.Counter(classify(op, s) for op, s in ins)
        tot.update(c)
        tgt = [s.split()[-1] for op, s in ins if op.startswith("s_cbranch") or op.startswith("s_branch")]
        print("%-12s %5d | %4d %4d %4d %4d %4d %4d %4d %4d %4d %4d %3d  -> %s" % (b, len(ins), c["mfma"], c["vpk"], c["vdpp"], c["valu"], c["vlane"],
              c["salu"], c["smem"], c["vmem"], c["lds"], c["wait"], c["br"], ",".join(tgt)))
        if dump:
            for op, s in ins: print("      ", s)
    print("total", dict(tot))

main()

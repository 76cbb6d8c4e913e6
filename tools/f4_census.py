#!/usr/bin/env python3
"""tools/f4_census.py — instruction census of k_kwtp16<38, EBU, REGPF> (layout 6) per tile and wave, by phase.

Static: the basic blocks a steady tile executes (48 kHz, len = 2400, mid-call, dense, no pruning) times their trip
counts, from hipcc's own assembly.  It reproduces the PMC counters of profiles/r02a (1356 VALU, 342 MFMA, 351 SALU,
124 LDS, 32 VMEM per tile and wave), i.e. the itemisation VERDICT r2 asked for.

    hipcc --offload-arch=gfx950 -O3 -std=c++20 -Iinclude -Imeters.lv2_amd/csrc -mllvm -amdgpu-mfma-vgpr-form \
          -fno-slp-vectorize -save-temps -c meters.lv2_amd/csrc/mtr_fused4.hip -o /tmp/f4.o      (in a scratch directory)
    python3 tools/f4_census.py mtr_fused4-hip-amdgcn-amd-amdhsa-gfx950.s

The block names are those of the compiler build the table was made with (ROCm 7.2, clang 22); a different compiler
renumbers them — re-derive the hot list with tools/isa_blocks.py --dump then.
"""
import collections, re, subprocess, sys, os

HOT = collections.OrderedDict([
 ("tile bookkeeping (tile_of, loop control, lane masks)", [(".LBB2_103",1),("%bb.104",1),(".LBB2_105",1),("%bb.108",1),(".LBB2_110",1),(".LBB2_112",1),(".LBB2_113",1),(".LBB2_115",1),(".LBB2_116",1),(".LBB2_119",1),(".LBB2_140",1),("%bb.141",1),("%bb.142",1),(".LBB2_143",1),(".LBB2_145",1),(".LBB2_147",1),(".LBB2_240",1),("%bb.277",1)]),
 ("scan row matrices (8 loads), max |x| per lane, exponent-pair reduction", [(".LBB2_107",1)]),
 ("split: scale, f16 hi / lo, LDS writes (+ the 24 halo lanes)", [("%bb.117",1),("%bb.118",1)]),
 ("K-filter pass 1 (end-state functionals, SGPR coefficients, carry into lane 0)", [(".LBB2_120",1),("%bb.122",1),("%bb.123",1),(".LBB2_124",1),("%bb.125",1)]),
 ("K-filter scan (DPP) + hand-over to the right neighbour", [(".LBB2_126",1),("%bb.127",1),("%bb.129",1),("%bb.130",1)]),
 ("K-filter pass 2 (kw_pair x 19, hand-scheduled)", [("%bb.131",1)]),
 ("tile power: wave sum, store, carried state picked from the last lane", [(".LBB2_135",1),("%bb.136",1),(".LBB2_137",1),(".LBB2_138",1)]),
 ("next tile's 19 loads + halo fetch", [("%bb.146",1),(".LBB2_225",1),("%bb.226",1),("%bb.227",1),(".LBB2_229",1),(".LBB2_230",1),(".LBB2_232",1),(".LBB2_236",1),(".LBB2_238",1),(".LBB2_239",1)]),
 ("products: 9 full blocks x 36 MFMA, operand reads, accumulator maxima", [("%bb.241",1),("%bb.242",1),("%bb.243",1),("%bb.259",1),("%bb.260",1),(".LBB2_261",9),(".LBB2_263",1),(".LBB2_272",1)]),
 ("products: the shared last block (18 MFMA) + masked maxima", [("%bb.273",1),(".LBB2_274",1),(".LBB2_275",1)]),
 ("peaks back to the samples' scale; run registers <- prefetch", [(".LBB2_276",1)]),
])

def main():
    here = os.path.dirname(os.path.abspath(__file__))
    dump = subprocess.run([sys.executable, os.path.join(here, "isa_blocks.py"), sys.argv[1], "k_kwtp16ILi38ELb1E"],
                          capture_output=True, text=True, check=True).stdout
    keys = "mfma vpk vdpp valu vlane salu smem vmem lds wait br".split()
    blocks = {}
    for l in dump.split("\n"):
        m = re.match(r"^(\S+)\s+(\d+) \|" + r"\s+(\d+)" * 11, l)
        if m: blocks[m.group(1)] = dict(zip(keys, map(int, m.groups()[2:])))
    # (the labels in HOT carry the function index of the build the table was made with: .LBB2_*; follow the current one)
    pref = next((b.split("_")[0] for b in blocks if b.startswith(".LBB")), ".LBB2")
    tot = collections.Counter()
    print("| phase | MFMA | packed f32 | DPP | other VALU | readlane | SALU | SMEM | VMEM | LDS | waitcnt / nop | branch |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, lst in HOT.items():
        c = collections.Counter()
        for b, n in lst:
            for k, v in blocks[b.replace(".LBB2", pref)].items(): c[k] += v * n
        tot.update(c)
        print("| " + name + " | " + " | ".join(str(c[k]) for k in keys) + " |")
    print("| **sum** | " + " | ".join(str(tot[k]) for k in keys) + " |")
    print("\nnon-MFMA VALU per tile and wave: %d   (PMC, profiles/r02a: 1356)" % (tot["vpk"] + tot["vdpp"] + tot["valu"] + tot["vlane"]))

main()

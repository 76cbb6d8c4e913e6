#!/usr/bin/env python3
"""tests/golden/make_golden_ttl.py — pin the generated LV2 metadata to the REFERENCE's own (build container only).

Parses /root/reference/lv2ttl/meters.lv2.ttl.in (the per-plugin port descriptions: VU :13-43, EBUr128 :614-660,
spectr30 :1292-1841, dBTP :1906-1975, ...) and manifest.lv2.ttl.in and writes, for every plugin URI, the tuples a host
that saved a session with the reference's bundle depends on — per port: index, symbol, port classes, and default /
minimum / maximum where the reference states them — to tests/golden/golden_ttl_v1.json.  tests/test_lv2_bundle.py holds
tools/gen_ttl.py's output against it.  Data only: no text of the reference's file is kept, the numbers and symbols are
the interface."""
import json
import os
import re
import sys

REF = "/root/reference/lv2ttl"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden_ttl_v1.json")


def ports_of(block):
    """The [ ... ] groups behind lv2:port, bracket-balanced."""
    i = block.find("lv2:port")
    if i < 0:
        return []
    out, depth, cur = [], 0, []
    for ch in block[i:]:
        if ch == "[":
            depth += 1
            if depth == 1:
                cur = []
                continue
        elif ch == "]":
            depth -= 1
            if depth == 0:
                out.append("".join(cur))
                continue
        if depth >= 1:
            cur.append(ch)
        elif ch == ";" and out:
            break
    return out


def num(txt, key):
    m = re.search(r"lv2:%s\s+(-?[0-9.]+(?:[eE][-+]?\d+)?)" % key, txt)
    return float(m.group(1)) if m else None


def main():
    ttl = open(os.path.join(REF, "meters.lv2.ttl.in")).read()
    man = open(os.path.join(REF, "manifest.lv2.ttl.in")).read()
    in_manifest = re.findall(r"^mtr:(\w+)@URI_SUFFIX@", man, flags=re.M)
    heads = [(m.group(1), m.start()) for m in re.finditer(r"^mtr:(\w+)@URI_SUFFIX@", ttl, flags=re.M)]
    plugins = {}
    for k, (name, pos) in enumerate(heads):
        block = ttl[pos:heads[k + 1][1] if k + 1 < len(heads) else len(ttl)]
        ports = []
        for p in ports_of(block):
            idx = re.search(r"lv2:index\s+(\d+)", p)
            sym = re.search(r'lv2:symbol\s+"([^"]+)"', p)
            if not idx or not sym:
                continue
            classes = sorted(set(re.findall(r"\b(?:lv2|atom):(ControlPort|AudioPort|AtomPort|InputPort|OutputPort|CVPort)\b", p)))
            ports.append({"index": int(idx.group(1)), "symbol": sym.group(1), "classes": classes,
                          "default": num(p, "default"), "minimum": num(p, "minimum"), "maximum": num(p, "maximum")})
        ports.sort(key=lambda q: q["index"])
        assert [q["index"] for q in ports] == list(range(len(ports))), name
        plugins[name] = {"ports": ports, "needs_urid_map": bool(re.search(r"lv2:requiredFeature[^;]*urid:map", block)),
                         "hard_rt_capable": "lv2:hardRTCapable" in block,
                         "min_atom_buffer": (int(re.search(r"rsz:minimumSize\s+(\d+)", block).group(1)) if "rsz:minimumSize" in block else None)}
    json.dump({"source": "x42/meters.lv2 v0.9.28 lv2ttl/meters.lv2.ttl.in + manifest.lv2.ttl.in, parsed by tests/golden/make_golden_ttl.py",
               "uri_prefix": "http://gareus.org/oss/lv2/meters#", "manifest": in_manifest, "plugins": plugins},
              open(OUT, "w"), indent=0, sort_keys=True)
    print("wrote", OUT, len(plugins), "plugins,", sum(len(v["ports"]) for v in plugins.values()), "ports")


if __name__ == "__main__":
    sys.exit(main())

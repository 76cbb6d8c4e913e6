"""The generated LV2 bundle metadata (tools/gen_ttl.py) against the REFERENCE's own metadata.

tests/golden/golden_ttl_v1.json holds, for every plugin of the reference's bundle, the tuples a host that saved a session
with that bundle depends on — per port: index, symbol, port classes, default / minimum / maximum — parsed from
lv2ttl/meters.lv2.ttl.in and manifest.lv2.ttl.in by tests/golden/make_golden_ttl.py in the build container (data, not the
file's text).  Every plugin gen_ttl.py declares must carry exactly those tuples: one changed symbol or index fails here
(round 5 compared the generator with literals typed into this file, and 333 tuples of the needle,
DR-14, TPnRMS and surround plugins were not the reference's)."""
import json
import os
import re
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
from make_golden_ttl import num, ports_of  # noqa: E402  (the parser that wrote the golden file reads the generated one)

GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "golden_ttl_v1.json")))
OUT_OF_SCOPE = {"goniometer", "phasewheel", "stereoscope"}         # GUI data pumps


def _generated(tmp_path, patch=None):
    src = open(os.path.join(ROOT, "tools", "gen_ttl.py")).read()
    if patch:
        assert src.count(patch[0]) >= 1, patch
        src = src.replace(patch[0], patch[1], 1)
    gen = tmp_path / "gen_ttl.py"
    gen.write_text(src)
    out = tmp_path / "bundle"
    subprocess.check_call([sys.executable, str(gen), str(out)], stdout=subprocess.DEVNULL)
    ttl = open(out / "meters_amd.ttl").read()
    man = open(out / "manifest.ttl").read()
    heads = [(m.group(1), m.start()) for m in re.finditer(r"^mtr:(\w+)$", ttl, flags=re.M)]
    plugins = {}
    for k, (name, pos) in enumerate(heads):
        block = ttl[pos:heads[k + 1][1] if k + 1 < len(heads) else len(ttl)]
        ports = []
        for p in ports_of(block):
            classes = sorted(set(re.findall(r"\b(?:lv2|atom):(ControlPort|AudioPort|AtomPort|InputPort|OutputPort|CVPort)\b", p)))
            ports.append({"index": int(re.search(r"lv2:index\s+(\d+)", p).group(1)), "symbol": re.search(r'lv2:symbol\s+"([^"]+)"', p).group(1),
                          "classes": classes, "default": num(p, "default"), "minimum": num(p, "minimum"), "maximum": num(p, "maximum")})
        plugins[name] = {"ports": ports, "needs_urid_map": bool(re.search(r"lv2:requiredFeature[^;]*urid:map", block)),
                         "hard_rt_capable": "lv2:hardRTCapable" in block,
                         "min_atom_buffer": (int(re.search(r"rsz:minimumSize\s+(\d+)", block).group(1)) if "rsz:minimumSize" in block else None)}
    return plugins, man


def _differences(plugins):
    diffs = []
    for name, got in plugins.items():
        ref = GOLD["plugins"].get(name)
        if ref is None:
            diffs.append((name, "not a plugin of the reference's bundle"))
            continue
        if len(got["ports"]) != len(ref["ports"]):
            diffs.append((name, "ports", len(got["ports"]), len(ref["ports"])))
            continue
        for a, b in zip(got["ports"], ref["ports"]):
            for key in ("index", "symbol", "classes", "default", "minimum", "maximum"):
                if a[key] != b[key]:
                    diffs.append((name, a["index"], key, a[key], b[key]))
        if got["needs_urid_map"] != ref["needs_urid_map"]:
            diffs.append((name, "urid:map", got["needs_urid_map"], ref["needs_urid_map"]))
        # the plugin may ask the host for a LARGER notify buffer than the reference does, never a smaller one
        if (ref["min_atom_buffer"] or 0) > (got["min_atom_buffer"] or 0):
            diffs.append((name, "rsz:minimumSize", got["min_atom_buffer"], ref["min_atom_buffer"]))
    return diffs


def test_ttl_is_the_references_interface(tmp_path):
    plugins, man = _generated(tmp_path)
    assert _differences(plugins) == []
    # every plugin of the reference's manifest that is in scope is declared, in the manifest and in the description, by its URI
    assert GOLD["uri_prefix"] == "http://gareus.org/oss/lv2/meters#" and "<%s>" % GOLD["uri_prefix"] in man
    want = [p for p in GOLD["manifest"] if p not in OUT_OF_SCOPE]
    assert sorted(plugins) == sorted(want) and len(want) == 33
    for p in want:
        assert f"mtr:{p}\n" in man
    assert man.count("lv2:binary <meters_amd.so>") == len(want)
    # the GPU plugins do not claim hard real-time capability (their run () takes driver locks), the host-CPU ones do as the reference
    for p in ("EBUr128", "dBTPstereo", "spectr30mono", "bitmeter", "dr14stereo"):
        assert not plugins[p]["hard_rt_capable"] and GOLD["plugins"][p]["hard_rt_capable"]
    for p in ("VUmono", "BBCstereo", "COR", "K14stereo", "surround5"):
        assert plugins[p]["hard_rt_capable"]


@pytest.mark.parametrize("patch", [('"levelM6"', '"levelM"'), ('"cor%dA"', '"cor%da"'), ('ctl(3, "level1", "Level", "Output", 0.0, 1.0)]', 'ctl(4, "level1", "Level", "Output", 0.0, 1.0)]'),
                                   ('"host_transport"', '"follow_transport"'), ('"max%d"', '"peak%d"')])
def test_one_changed_symbol_or_index_fails(tmp_path, patch):
    """The check has teeth: a generator with ONE symbol or index off is caught."""
    try:
        plugins, _ = _generated(tmp_path, patch)
    except (AssertionError, subprocess.CalledProcessError):
        raise
    assert _differences(plugins) != []


def test_the_plugin_serves_every_uri_the_bundle_declares():
    """lv2_descriptor () of meters_amd.so enumerates at least the URIs of the generated manifest (no GPU needed: nothing is instantiated)."""
    import ctypes as C
    so = os.path.join(ROOT, "meters.lv2_amd", "lib", "meters_amd.so")
    if not os.path.exists(so):
        pytest.skip("meters_amd.so not built")
    lib = C.CDLL(so)

    class Desc(C.Structure):
        _fields_ = [("URI", C.c_char_p)] + [(n, C.c_void_p) for n in ("instantiate", "connect_port", "activate", "run", "deactivate", "cleanup", "extension_data")]
    lib.lv2_descriptor.restype = C.POINTER(Desc)
    lib.lv2_descriptor.argtypes = [C.c_uint32]
    uris, i = set(), 0
    while True:
        d = lib.lv2_descriptor(i)
        if not d:
            break
        uris.add(d.contents.URI.decode())
        i += 1
    for p in GOLD["manifest"]:
        if p not in OUT_OF_SCOPE:
            assert GOLD["uri_prefix"] + p in uris, p

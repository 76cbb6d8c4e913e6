"""The generated LV2 bundle metadata (tools/gen_ttl.py) must describe exactly the ports the plugin's
connect_port() wires (SURVEY.md §8b port maps)."""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ttl_matches_the_plugin_port_maps(tmp_path):
    subprocess.check_call([sys.executable, os.path.join(ROOT, "tools", "gen_ttl.py"), str(tmp_path)])
    man = open(tmp_path / "manifest.ttl").read()
    ttl = open(tmp_path / "meters_amd.ttl").read()
    plugs = ["VUmono", "VUstereo", "EBUr128", "spectr30mono", "dBTPmono", "dBTPstereo", "spectr30stereo",
             "SigDistHist", "bitmeter"]
    for p in plugs:
        assert f"mtr:{p}\n" in man and "lv2:binary <meters_amd.so>" in man
    blocks = {p: ttl.split(f"mtr:{p}\n")[1].split("\n\t.\n")[0] for p in plugs}
    want = {"VUmono": 4, "VUstereo": 7, "EBUr128": 6, "spectr30mono": 66, "dBTPmono": 5, "dBTPstereo": 9,
            "spectr30stereo": 68, "SigDistHist": 4, "bitmeter": 4}
    needles = {"BBCmono": 4, "EBUstereo": 7, "DINmono": 4, "NORstereo": 7, "COR": 6, "BBCM6": 8, "K14mono": 6, "K20stereo": 10,
               "dr14mono": 11, "dr14stereo": 19, "TPnRMSmono": 11, "TPnRMSstereo": 19,
               "surround3": 25, "surround5": 33, "surround8": 45}
    for p, n in needles.items():
        assert f"mtr:{p}\n" in man
        blk = ttl.split(f"mtr:{p}\n")[1].split("\n\t.\n")[0]
        assert [int(i) for i in re.findall(r"lv2:index (\d+)", blk)] == list(range(n)), p
    for p in ("SigDistHist", "bitmeter"):
        assert re.findall(r'lv2:symbol "([^"]+)"', blocks[p]) == ["control", "notify", "in", "out"]
        assert "lv2:requiredFeature urid:map" in blocks[p]
    for p, n in want.items():
        idx = [int(i) for i in re.findall(r"lv2:index (\d+)", blocks[p])]
        assert idx == list(range(n)), p
    sym = re.findall(r'lv2:symbol "([^"]+)"', blocks["spectr30stereo"])
    assert sym[0] == "band25" and sym[29] == "band20000" and sym[30] == "max25" and sym[59] == "max20000"
    assert sym[60:] == ["UIspeed", "UIreset", "UIgain", "UImiscstate", "inL", "outL", "inR", "outR"]
    assert re.findall(r'lv2:symbol "([^"]+)"', blocks["dBTPstereo"]) == \
        ["ref", "inL", "outL", "levelL", "inR", "outR", "levelR", "peakL", "peakR"]
    assert re.findall(r'lv2:symbol "([^"]+)"', blocks["EBUr128"]) == ["control", "notify", "inL", "outL", "inR", "outR"]
    assert "lv2:requiredFeature urid:map" in blocks["EBUr128"] and "rsz:minimumSize 4096" in blocks["EBUr128"]

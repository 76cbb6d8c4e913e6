import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline", {})
print(sys.argv[1:], "step %.2f ms" % d["ms_per_step"], "kernel %.2f ms" % r.get("kernel_ms", 0),
      "%.1f GB/s" % r.get("achieved", 0), "frac %.3f" % r.get("frac", 0), d.get("check"))

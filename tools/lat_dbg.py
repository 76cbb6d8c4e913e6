import sys, time, numpy as np
sys.path.insert(0, "tests"); sys.path.insert(0, ".")
from _lv2host import Host
import _lv2lat as L
host = Host()
for n in (64, 256):
    keep = {"ui": False}
    inst, hook = L._wire(host, "EBUr128", n, keep)
    t = []
    for i in range(1200):
        if hook: hook()
        t0 = time.perf_counter(); inst.run(n); t.append(time.perf_counter() - t0)
    t = np.array(t) * 1e6
    idx = np.argsort(t)[-12:]
    print(n, "median", np.median(t), "slow:", sorted([(int(i), int(i * n // 2400), round(float(t[i]), 1)) for i in idx]))
    inst.cleanup()

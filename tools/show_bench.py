import json, sys
d = json.loads([l for l in open(sys.argv[1]).read().strip().splitlines() if l.startswith("{")][-1])
print("ms/step", round(d["ms_per_step"], 3), "value %.3e" % d["value"])
for k, v in d["kernels"].items():
    print("  %-18s n=%-3d avg_ms=%.4f alg_GB/s=%s" % (k, v["launches"], v["avg_ms"], round(v["alg_GBps"]) if v["alg_GBps"] else None))

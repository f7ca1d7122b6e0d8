"""prints the headline fields of bench.py JSON lines: python tools/show_bench.py <file> [<file> ...]"""
import json
import sys

for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    r, c = d.get("roofline") or {}, d.get("cpu_baseline") or {}
    print("%s: %.0f %s, %.4f ms/step, n_gpus %s, frac %s, kernel_ms %s, cpu %s, host_pointer_qps %s" % (
        f, d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], r.get("frac"), r.get("kernel_ms"), c.get("value"), d.get("host_pointer_qps")))

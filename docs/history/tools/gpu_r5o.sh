#!/bin/bash
# round 5, last state: the whole GPU suite + smoke, then the default bench line (with the 20-state call_latency rows)
export TMPDIR=/tmp
o=gpurun_out/r5o; mkdir -p $o
timeout 1800 python -m pytest tests -q -m gpu > $o/full.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" $o/full.log | tail -2; grep -E "^FAILED|^ERROR" $o/full.log | head -20
timeout 120 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > $o/bench_default.json 2> $o/bench_default.err; echo "bench rc=$?"
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5o/bench_default.json").read().strip().splitlines()[-1])
print("cfg2", d["value"], d["ms_per_step"], d["roofline"].get("kernel_avg_us"), d["roofline"].get("frac_real"))
for k, v in d["extra"]["call_latency"].items():
    if isinstance(v, dict) and "us_per_candidate" in v:
        print(k, round(v["us_per_candidate"], 2), round(v["us_per_scalar_returning_call"], 2), v["served_by_resident_workgroups"])
PY

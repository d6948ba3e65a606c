python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py --steps 12 --no-secondary --no-pcie --no-cpu-baseline 2>/dev/null > /tmp/b.json
python3 - <<'P'
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print(round(d["value"],3), d["roofline"]["frac"], d["stage_ms_per_step"], d.get("lp_worker_utilisation"))
P

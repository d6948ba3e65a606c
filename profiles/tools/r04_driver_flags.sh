for k in 1 2 3; do
timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-secondary --no-pcie --no-cpu-baseline 2>/dev/null > /tmp/b.json
python3 - <<'P'
import json
d=json.loads(open('/tmp/b.json').read().strip().splitlines()[-1])
print(round(d["value"],3), round(d["whole_stream_value"],3), d["roofline"]["frac"], d.get("lp_worker_utilisation"), d["host_s_per_step"]["lp"], d["timed_region"]["timed_s"], d["host_s_per_step"]["gpu_thread"], d["host_s_per_step"]["intervals"])
P
done

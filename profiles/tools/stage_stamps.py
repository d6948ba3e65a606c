#!/usr/bin/env python3
"""Summarise DALIGN_DEBUG_TIMES=1 stamps of da_pair_stage (stderr of bench.py): per stage the wall time, the wait for the GEMM,
the result slot, and the idle time of the feeding thread between two stages.  usage: stage_stamps.py <stderr file> [first=60]"""
import re
import sys

rows, cur = [], {}
for line in open(sys.argv[1], errors="replace"):
  m = re.search(r"\[match_finish\] GEMM done.*\+\s*([\d.]+) ms", line)
  if m:
    cur["gemm_wait"] = float(m.group(1))
  m = re.search(r"kernels: features ([\d.]+) prep ([\d.]+) gemm ([\d.]+) verify ([\d.]+).*result slot (-?\d+)", line)
  if m:
    cur.update(gemm=float(m.group(3)), slot=int(m.group(5)))
  m = re.search(r"\[pair_stage\] total\s+([\d.]+) ms \(entered at ([\d.]+) ms\)", line)
  if m:
    cur.update(total=float(m.group(1)), entered=float(m.group(2)))
    rows.append(cur); cur = {}
first = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rows = rows[first:]
for a, b in zip(rows, rows[1:]):
  a["idle_after"] = b["entered"] - a["entered"] - a["total"]
rows = rows[:-1]
def stat(key, sel=lambda r: True):
  v = sorted(r[key] for r in rows if key in r and sel(r))
  return "n=%d p10=%.2f p50=%.2f p90=%.2f max=%.2f" % (len(v), v[len(v) // 10], v[len(v) // 2], v[len(v) * 9 // 10], v[-1]) if v else "n=0"
print("stages", len(rows))
print("stage wall      ", stat("total"))
print("  slot 0        ", stat("total", lambda r: r.get("slot") == 0))
print("  slot 1        ", stat("total", lambda r: r.get("slot") == 1))
print("wait for GEMM   ", stat("gemm_wait"))
print("GEMM (events)   ", stat("gemm"))
print("idle after stage", stat("idle_after"))
late = [r for r in rows if r.get("gemm_wait", 0) - r.get("gemm", 0) > 3]
print("stages whose GEMM started > 3 ms late:", len(late), "of", len(rows), "; by slot:", {s: sum(1 for r in late if r.get("slot") == s) for s in (0, 1, 2, 3)})

# where the long stages lose their time: per stage the four waits of da_match_finish (stamps in order)
import collections
segs = collections.defaultdict(list)
cur = []
for line in open(sys.argv[1], errors="replace"):
  m = re.search(r"\[match_finish\] (.*?)\s+\+\s*([\d.]+) ms", line)
  if m:
    cur.append((m.group(1).strip(), float(m.group(2))))
  m = re.search(r"\[pair_stage\] total\s+([\d.]+) ms", line)
  if m:
    segs[float(m.group(1)) > 48.0].append(cur); cur = []
for long_, rows_ in segs.items():
  agg = collections.defaultdict(list)
  for r in rows_[first if not long_ else 0:]:
    for name, v in r:
      agg[name].append(v)
  print("LONG stages (> 48 ms):" if long_ else "normal stages:", len(rows_), {k: round(sum(v) / len(v), 2) for k, v in agg.items()})

#!/usr/bin/env python3
"""What happens on the device between two similarity GEMMs of a batch?  Reads a rocprofv3 --kernel-trace CSV of
`bench.py --workload cfgX` and prints, over the steady-state launches: the GEMM's mean duration, the mean gap from the end
of one GEMM to the start of the next, and the kernels that ran inside those gaps (mean busy time per gap, by name).
usage: gemm_gap_trace.py <kernel_trace.csv> [skip_first=40]"""
import collections
import csv
import json
import sys


def main():
  path = sys.argv[1]
  skip = int(sys.argv[2]) if len(sys.argv) > 2 else 40
  rows = []
  for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", "").replace("da::", "")))
  rows.sort()
  gemms = [r for r in rows if r[2].startswith("k_match_")]
  gemms = gemms[skip:-8] if len(gemms) > skip + 16 else gemms
  dur = [e - s for s, e, _ in gemms]
  gaps, inside = [], collections.defaultdict(float)
  k = 0
  for (s0, e0, _), (s1, e1, _) in zip(gemms, gemms[1:]):
    gaps.append(s1 - e0)
    while k < len(rows) and rows[k][1] <= e0:
      k += 1
    j = k
    while j < len(rows) and rows[j][0] < s1:
      s, e, n = rows[j]
      if not n.startswith("k_match_"):
        inside[n] += max(0, min(e, s1) - max(s, e0))
      j += 1
  n = max(1, len(gaps))
  out = dict(trace=path, gemm_launches=len(gemms), gemm_ms_mean=round(sum(dur) / max(1, len(dur)) * 1e-6, 3),
             gap_ms_mean=round(sum(gaps) / n * 1e-6, 3), period_ms_mean=round((gemms[-1][0] - gemms[0][0]) / n * 1e-6, 3),
             busy_ms_per_gap_by_kernel={a: round(b / n * 1e-6, 3) for a, b in sorted(inside.items(), key=lambda x: -x[1])[:24]})
  # the timeline of one typical gap (the median one): every kernel that overlaps it, times relative to the GEMM's end
  order = sorted(range(len(gaps)), key=lambda t: gaps[t])
  t = order[len(order) // 2]
  e0, s1 = gemms[t][1], gemms[t + 1][0]
  line = []
  for s_, e_, n_ in rows:
    if e_ > e0 and s_ < s1 and not n_.startswith("k_match_"):
      line.append([round((s_ - e0) * 1e-3, 1), round((e_ - e0) * 1e-3, 1), n_[:60]])
  out["median_gap_timeline_us"] = dict(gap_us=round((s1 - e0) * 1e-3, 1), kernels=line[:160])
  print(json.dumps(out, indent=1))


if __name__ == "__main__":
  main()

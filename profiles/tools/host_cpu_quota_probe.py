"""Host probe: is the box's CPU capacity what `nproc` says?  W processes each run the same fixed pure-CPU loop (no memory
traffic beyond L1); if the wall time grows with W long before W reaches the number of cores, the container runs under a
CPU quota (cgroup cpu.max) or shares its cores, and the 'host LP capacity' figures are that quota's, not the caches'."""
import json, multiprocessing as mp, os, time


def spin(args):
  cpu, t_start = args
  if cpu is not None:
    try:
      os.sched_setaffinity(0, {cpu})
    except Exception:
      pass
  while time.time() < t_start:
    pass
  t0 = time.time(); c0 = time.process_time()
  x = 0
  for i in range(6_000_000):
    x += i * i & 7
  return time.time() - t0, time.process_time() - c0


def main():
  for name in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu.stat",
               "/sys/fs/cgroup/cpuset.cpus.effective"):
    try:
      print(name, "=", open(name).read().strip().replace("\n", " | "))
    except Exception as e:
      print(name, "unreadable:", type(e).__name__)
  print("sched_getaffinity:", len(os.sched_getaffinity(0)), "cpu_count:", os.cpu_count(), "loadavg:", open("/proc/loadavg").read().strip())
  import sys
  sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
  from describealign_amd import align as A
  order = A.cpu_order()
  for pinned in (True, False):
    for w in (1, 8, 16, 32, 64, 128):
      t_start = time.time() + 1.0 + 0.01 * w
      with mp.get_context("fork").Pool(w) as pool:
        res = pool.map(spin, [((order[k % len(order)] if pinned else None), t_start) for k in range(w)], chunksize=1)
      wall = [r[0] for r in res]; cpu = [r[1] for r in res]
      print(json.dumps(dict(pinned=pinned, workers=w, wall_mean_s=round(sum(wall) / w, 3), wall_max_s=round(max(wall), 3),
                            cpu_mean_s=round(sum(cpu) / w, 3))), flush=True)


if __name__ == "__main__":
  main()

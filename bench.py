#!/usr/bin/env python3
"""Benchmark of the alignment hot path on MI355X.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload cfg2|cfg1|cfg-small] [--precision f32|bf16]

One step = one pass of the hot path over one (video, AD) pair whose PCM is already resident in
HBM: feature kernel (both sides) -> similarity GEMM + verification -> chain DP (device) -> host
LP -> banded extension + second DP -> nodes.  metric = aligned audio-hours/s (video-side duration
of the pairs processed / wall time), whole job over all ranks.  With N > 1 (launched by
torch.distributed.run, one rank per GPU) every rank aligns its own pairs: a directory batch shards
with no data-path collective (weak scaling).  The stream rotates through --distinct (4) different pairs
of the workload's shape per rank (seeds 5 + rank, ...), all resident in HBM: consecutive steps never
align the same pair, the LP and the survivor counts are means over the rotation.

Workloads (BASELINE.json configs):
  cfg2  configs[2]: 7200 s stereo pair, 10 jumps, bf16 MFMA GEMM -- the largest single-GPU
        configuration and the north_star target                                      (default)
  cfg1  configs[1] stand-in: 1320 s video / ~1558 s AD, 10 jumps + 200 s intro, mono, fp32 GEMM
        (reported as the `secondary` object of the same JSON line at N = 1)
  cfg3  configs[3], one GPU's share: 1800 s mono pairs, bf16 prefilter GEMM (--gpus 8 runs the whole
        config; reported as the `secondary_cfg3` object of the same JSON line at N = 1)

Timed region (steady state of a directory batch).  The pairs go through ONE primed pipeline: an
untimed lead-in of max(W, two LP solves per host worker) pairs, K timed pairs and a tail that keeps
the pipeline full are submitted as one stream; the clock runs from the completion of the lead-in's last
pair to the completion of K more pairs (completion times, not in-order delivery times: one slow LP solve
holds back the delivery of the finished pairs behind it and then releases them in a burst).  (The lead-in is
longer than --warmup because a pair spends ~2 s in the host LP but only ~0.17 s on the GPU: the first
results of a fresh pipeline come out in a burst at the GPU stage's rate, before the LP stage has had
to sustain anything; `--prime 0` disables it.)  A barrier and a
device synchronisation bracket the run of the stream (before the first job is submitted, after the
last result); inside it the K timed results are complete on the host when the clock stops, so no
work of the timed steps is left out -- what is NOT waited for at that instant is the in-flight
work of the tail pairs, which is not part of the K steps.  elapsed = max over ranks.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOADS = {
  "cfg1": dict(desc="configs[1] stand-in: synthetic 1320 s video / ~1558 s AD, 10 jumps + 200 s intro, mono",
               seconds=1320.0, n_jumps=10, first_gap=200.0, channels=1, precision="f32"),
  "cfg2": dict(desc="configs[2]: synthetic 7200 s (2 h) stereo pair, 10 injected offset jumps + 200 s intro",
               seconds=7200.0, n_jumps=10, first_gap=200.0, channels=2, precision="bf16"),
  "cfg3": dict(desc="configs[3] per-GPU share: synthetic 1800 s (30 min) mono pairs, 10 jumps + 120 s intro (the batch of 32 shards "
                    "embarrassingly: every rank streams its own pairs)",
               seconds=1800.0, n_jumps=10, first_gap=120.0, channels=1, precision="bf16"),
  "cfg-small": dict(desc="600 s mono pair, 5 jumps (CI-sized)",
                    seconds=600.0, n_jumps=5, first_gap=60.0, channels=1, precision="f32"),
}
PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}     # dense MFMA peaks, MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0


def cpu_baseline(channels, workload_seconds, sample_seconds=None, n_jumps=10, first_gap=200.0, golden_case=None):
  """The oracle (numpy restatement of the reference algorithm) on ONE host core.  The STATED value is the rate at the
  workload's own duration: measured when the sample IS the workload (configs[1]: 1320 s, ~10 s of CPU), otherwise the bounded
  sample (configs[2]: 2400 s of the 7200 s stereo pair, ~100 s of CPU) extrapolated -- features linearly, align() quadratically
  (stage 2 of the reference is quadratic in the duration).  The sample's own rate (an UPPER bound for the
  workload's) is kept as `sample_value`."""
  from describealign_amd import synth
  from oracle import dalign_oracle as O
  if sample_seconds is None:
    # the whole workload when that is ~10 s of CPU (configs[1]); else a third of it: 2 400 s of the 2 h stereo pair is ~100 s on
    # one core of the GPU box's host (3 600 s measured: 230 s, profiles/r06_bench_default_first.json) and is extrapolated x 9
    sample_seconds = workload_seconds if workload_seconds <= 1500.0 else min(workload_seconds, 2400.0)
  full = sample_seconds >= workload_seconds
  sample_seconds = min(sample_seconds, workload_seconds)
  n_j = n_jumps if full else max(2, int(round(n_jumps * sample_seconds / workload_seconds)))
  pair = synth.make_pair(3, sample_seconds, n_jumps=n_j, first_gap=first_gap if full else min(first_gap, 60.0), channels=channels)
  t0 = time.perf_counter()
  vf, af = O.features(pair.video), O.features(pair.audio)
  t1 = time.perf_counter()
  x, y, sim, path, med = O.align(vf, af, vf[0], af[0])
  t2 = time.perf_counter()
  dt = t2 - t0
  ratio = workload_seconds / sample_seconds
  est = (t1 - t0) * ratio + (t2 - t1) * ratio * ratio
  try:
    model = [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0]
  except Exception:
    model = "unknown"
  out = dict(
      value=(workload_seconds / 3600.0) / est, unit="audio-hours/s", cores=1, kind="port",
      value_is="measured at the workload's duration" if full else "extrapolated from the sample to the workload's duration",
      note="the numpy restatement of the reference algorithm (oracle/dalign_oracle.py), about 7x faster than the reference's own "
           "Python (describealign.py measured in the survey container: 65.6 s for a 1320 s pair, 0.0056 audio-hours/s); one core, "
           "as the reference is single-threaded.  `value` is the rate AT THE WORKLOAD'S DURATION; `sample_value` is the bounded "
           "sample's own rate, an upper bound for it",
      sample=f"{sample_seconds:.0f} s video / {pair.audio_seconds:.0f} s AD synthetic {'stereo' if channels == 2 else 'mono'} pair, "
             f"{n_j} jumps; features {t1 - t0:.1f} s + align {t2 - t1:.1f} s through oracle/dalign_oracle.py on one core",
      sample_value=(sample_seconds / 3600.0) / dt, sample_seconds_of_audio=sample_seconds,
      seconds=round(dt, 2), workload_seconds_estimated=round(est, 1), host_cpu=model, host_cores=os.cpu_count(),
      extrapolation=None if full else f"features x{ratio:.1f} (linear) + align x{ratio * ratio:.0f} (stage 2 is quadratic in the duration)")
  if golden_case:
    # what the REFERENCE ITSELF took on this very pair when its fixture was recorded (build container, one core)
    try:
      idx = json.load(open(os.path.join(ROOT, "tests", "golden", "index.json")))
      ref_s = idx["align"][golden_case].get("seconds_reference_total")
      if ref_s:
        out["reference_python_build_container_s"] = ref_s
        out["reference_python_build_container_value"] = (workload_seconds / 3600.0) / ref_s
        out["reference_python_note"] = (f"describealign.py v2.0.8 run on this pair (tests/golden/make_golden.py {golden_case}, features + align, "
                                        "in the build container, not on this host)")
    except Exception:
      pass
  return pair, (x, y), out


class Bench:
  def __init__(self, args, grp, device):
    self.args, self.grp, self.device = args, grp, device
    self.pairs = {}

  def sync(self):
    import torch
    torch.cuda.synchronize()
    self.grp.barrier()
    torch.cuda.synchronize()

  def run(self, workload, steps, warmup, with_cpu_baseline, with_stretch, include_h2d=False, quick=False):
    import contextlib, io, threading
    from describealign_amd import _native, synth
    from describealign_amd import align as A
    args, grp = self.args, self.grp
    rank, world = grp.rank, grp.world
    wl = WORKLOADS[workload]
    prec_name = args.precision or wl["precision"]
    prec = _native.PREC_F32 if prec_name == "f32" else _native.PREC_BF16
    workers = args.pipeline
    if workers < 0:
      workers = A.default_worker_count(int(os.environ.get("LOCAL_WORLD_SIZE", world)))
    ctx = _native.Context(self.device, prec)
    # every rank aligns its own pair (seed differs per rank): a sharded directory batch
    if workload not in self.pairs:
      self.pairs[workload] = synth.make_pair(5 + rank, wl["seconds"], n_jumps=wl["n_jumps"], first_gap=wl["first_gap"], channels=wl["channels"])
    pair = self.pairs[workload]
    gpu_ctxs = [ctx] + [_native.Context(self.device, prec) for _ in range(max(1, args.gpu_streams) - 1 if workers > 0 else 0)]
    include_h2d = include_h2d or args.include_h2d
    for c in gpu_ctxs:                       # the PCM is resident in HBM before anything is timed
      if include_h2d:
        # PCIe-inclusive mode: the PCM sits in page-locked host memory, where a decoder would have
        # written it (da_host_alloc); every step re-uploads it asynchronously (da_pcm_upload_async)
        if not hasattr(self, "pinned"):
          self.pinned = {}
        if workload not in self.pinned:
          pv = _native.pinned_empty(pair.video.shape); pv[...] = pair.video
          pa = _native.pinned_empty(pair.audio.shape); pa[...] = pair.audio
          self.pinned[workload] = (pv, pa)
        c.pcm_upload_async(_native.SIDE_VIDEO, self.pinned[workload][0])
        c.pcm_upload_async(_native.SIDE_AUDIO, self.pinned[workload][1])
      else:
        c.pcm_upload(_native.SIDE_VIDEO, pair.video)
        c.pcm_upload(_native.SIDE_AUDIO, pair.audio)
    h2d_ms = ctx.stats()["h2d_ms"]
    h2d_acc = []
    feat = {}
    lock = threading.Lock()

    fused = workers > 0 and not args.separate_calls

    # Distinct pairs in the stream.  The context holds this rank's pair; `distinct - 1` more pairs of the same shape (the next
    # seeds) wait in device buffers of their own (PcmStream), and every job adopts the next of those buffers -- a pointer swap,
    # the context's previous PCM goes into the stream object -- so that consecutive steps align DIFFERENT pairs, all resident in
    # HBM, in rotation: the LP and the survivor counts differ from pair to pair, the figures are means over the rotation.
    distinct = max(1, args.distinct) if (fused and not include_h2d and len(gpu_ctxs) == 1) else 1
    metas = [synth.SynthPair(video=np.empty((wl["channels"], 0), np.int16), audio=np.empty((wl["channels"], 0), np.int16),
                             jump_video_times=pair.jump_video_times, jump_lengths=pair.jump_lengths, seed=pair.seed)]
    rot = []
    held = {"ctx": 0, "slots": list(range(1, distinct))}       # which pair the context / each stream slot holds right now
    which = {}                                                   # job index -> pair it aligned

    def as_stream(pcm):
      st = _native.PcmStream(self.device, wl["channels"], pcm.shape[1])
      frames = np.ascontiguousarray(pcm.T)                    # interleaved (n, C), as a decoder delivers it
      for at in range(0, len(frames), 1 << 24):
        st.piece(frames[at:at + (1 << 24)])
      st.sync()
      return st

    def bring(c, p):
      """Make pair p the one resident in the context (da_pcm_exchange: swap with the slot that holds it, nothing is copied)."""
      if held["ctx"] != p:
        s_ = held["slots"].index(p)
        c.pcm_exchange(_native.SIDE_VIDEO, rot[s_][0]); c.pcm_exchange(_native.SIDE_AUDIO, rot[s_][1])
        held["ctx"], held["slots"][s_] = p, held["ctx"]

    if distinct > 1:
      # the context's own pair has to be interleaved too (exchange swaps like with like): it goes in through a stream as well
      for side, pcm in ((_native.SIDE_VIDEO, pair.video), (_native.SIDE_AUDIO, pair.audio)):
        st0 = as_stream(pcm)
        ctx.pcm_adopt(side, st0)
        st0.close()
    for j in range(1, distinct):
      other = synth.make_pair(5 + rank + j, wl["seconds"], n_jumps=wl["n_jumps"], first_gap=wl["first_gap"], channels=wl["channels"])
      rot.append([as_stream(other.video), as_stream(other.audio)])
      metas.append(synth.SynthPair(video=np.empty((wl["channels"], 0), np.int16), audio=np.empty((wl["channels"], 0), np.int16),
                                   jump_video_times=other.jump_video_times, jump_lengths=other.jump_lengths, seed=other.seed))
      del other

    def make_job(idx):
      def job(c):
        if include_h2d:
          c.pcm_upload_async(_native.SIDE_VIDEO, self.pinned[workload][0])
          c.pcm_upload_async(_native.SIDE_AUDIO, self.pinned[workload][1])
        if fused and not include_h2d:
          # the pair's PCM is resident: the pipeline runs features + matching + chain enqueue as ONE native call (da_pair_stage);
          # the feature times come back with that call's statistics (tm["device"])
          if distinct > 1:
            bring(c, idx % distinct)                 # jobs run one after the other on the context's own thread
          which[idx] = held["ctx"]
          return A.RESIDENT_PCM
        vf = c.features_resident(_native.SIDE_VIDEO)
        s_v = c.stats()
        af = c.features_resident(_native.SIDE_AUDIO)
        s_a = c.stats()
        with lock:
          feat[idx] = (s_v["features_ms"] + s_a["features_ms"], s_v["features_bytes"] + s_a["features_bytes"])
          if include_h2d:
            h2d_acc.append(s_v["h2d_ms"] + s_a["h2d_ms"])
        return vf, af
      return job

    quiet = contextlib.redirect_stdout(io.StringIO())
    tms, outs = [], []
    # tail: pairs submitted behind the timed ones so that every stage is as busy when the clock stops as
    # in the middle of a long batch: the LP solves of the timed pairs must run beside a full set of
    # other solves (they share the container's CPU-time quota: alone a 2 h pair's LP takes 1.7-2.4 s, beside 19 others
    # 1.8-3.4 s), so one whole generation of solves follows the last timed pair
    tail = 0 if workers <= 0 else (args.tail if args.tail >= 0 else (workers if not quick else workers // 3))
    # Priming.  A pair spends seconds in the host LP stage (2 h pair: ~2 s), far longer than the
    # 0.3 s between pairs, so the first results of a fresh pipeline arrive in a burst at the GPU
    # stage's rate: every worker is still on its FIRST solve and the LP stage has not yet had to
    # keep up.  The untimed lead-in is therefore at least `prime` pairs (two solves per worker by
    # default), so that the timed pairs see the rate a long directory batch sustains.
    declared_warmup = warmup
    if workers > 0:
      prime = args.prime if args.prime >= 0 else (2 * workers if not quick else workers // 2)
      warmup = max(warmup, prime)
    total = warmup + steps + tail
    pipe = None
    if workers > 0:
      pipe = A.AlignPipeline(gpu_ctxs, lp_workers=workers)
      pipe.warm()
    self.sync()
    t_start = time.perf_counter()
    t0 = t_start if warmup == 0 else None
    t1 = None
    with quiet:
      if pipe is not None:
        it = pipe.run((make_job(k) for k in range(total)), timings=tms, expected=total)
        for k in range(total):
          o = next(it)
          outs.append(o if k == warmup + steps - 1 else None)          # a 2 h pair's path is ~60 MB
          now = time.perf_counter()
          if k == warmup - 1:
            t0 = now
          if k == warmup + steps - 1:
            t1 = now
      else:
        for k in range(total):
          vf, af = make_job(k)(ctx)
          tm = {}
          o = A.align(vf, af, vf[0], af[0], ctx=ctx, timings=tm)
          outs.append(o if k == warmup + steps - 1 else None)
          tms.append(tm)
          now = time.perf_counter()
          if k == warmup - 1:
            t0 = now
          if k == warmup + steps - 1:
            t1 = now
    self.sync()
    t_end = time.perf_counter()
    if pipe is not None:
      # The clock is read on COMPLETION times, not on delivery: results are handed back in submission
      # order, so one slow LP solve holds back the finished pairs behind it and releases them in a
      # burst -- over 20 pairs that is +-25 % of noise.  t0 = the moment the lead-in's last pair
      # completed, t1 = the moment K more pairs had completed (whichever pairs those were).
      done_times = sorted(tm["done_t"] for tm in tms)
      t0 = done_times[warmup - 1] if warmup > 0 else t_start
      t1 = done_times[warmup + steps - 1]
    tms_by_idx = {k: tm for k, tm in enumerate(tms)}          # results are delivered in submission order
    elapsed = grp.max_over_ranks(t1 - t0)
    out = outs[warmup + steps - 1]
    sel = tms[warmup:warmup + steps]
    acc = {}

    def add(k, v):
      acc[k] = acc.get(k, 0.0) + v

    for tm in sel:
      d = tm["device"]
      for k in ("gemm_ms", "gemm_flops", "gemm_pairs", "verify_ms", "verify_kernel_ms", "prep_ms", "chain_ms", "refine_kernel_ms",
                "refine_dp_ms", "survivors", "matches"):
        add(k, d[k])
      add("lp_s", tm["lp_s"]); add("match_s", tm["match_s"]); add("chain_s", tm["chain_s"])
      add("n_path1", tm["n_path1"]); add("n_fit_points", tm["n_fit_points"])
      add("align_s", tm["match_s"] + tm["chain_s"] + tm["pass1_host_s"] + tm["lp_s"] + tm["cluster_s"] + tm["refine_s"] + tm["nodes_s"])
      for name in ("worker_cpu_s", "gpu_thread_cpu_s", "refine_cpu_s", "handoff_cpu_s"):
        add(name, tm.get(name, 0.0))
      for name in ("pass1_host_s", "cluster_s", "refine_s", "nodes_s", "worker_s", "features_s", "pace_s", "chain_begin_s",
                   "match_begin_s", "collect_under_gemm_s", "match_finish_s"):
        add(name, tm.get(name, 0.0))
      if "t_handoff" in tm and "done_t" in tm:      # where a pair spends its time between the stages (pipelined runs)
        add("iv_copy", tm["t_copied"] - tm["t_handoff"]); add("iv_wait_for_worker_slot", tm["t_submitted"] - tm["t_copied"])
        add("iv_in_worker_pool", tm["t_worker_back"] - tm["t_submitted"]); add("iv_wait_for_refine_thread", tm["t_refine_start"] - tm["t_worker_back"])
        add("iv_refine_and_nodes", tm["done_t"] - tm["t_refine_start"])
    for idx in range(warmup, warmup + steps):
      if idx in feat:
        add("feat_ms", feat[idx][0]); add("feat_bytes", feat[idx][1])
      else:                                    # fused stage: both sides' feature kernels are in the pair's own statistics
        add("feat_ms", tms_by_idx[idx]["device"]["features_ms"]); add("feat_bytes", tms_by_idx[idx]["device"]["features_bytes"])

    # accuracy of the recovered piecewise offsets against the injected truth (the pair the last timed step aligned)
    x, y = out[0], out[1]
    last_pair = which.get(warmup + steps - 1, 0)
    inj_err_ms = _offset_error_ms(metas[last_pair], x, y)
    if distinct > 1:
      bring(ctx, last_pair)                           # the auxiliary measurements below run on the PCM `out` belongs to

    res = None
    if rank == 0:
      k = float(steps)
      hours = wl["seconds"] / 3600.0
      value = hours * world * steps / elapsed
      gemm_tf = acc["gemm_flops"] / (acc["gemm_ms"] * 1e-3) / 1e12 if acc.get("gemm_ms") else 0.0
      peak = PEAK_TFLOPS[prec_name]
      # the same figure over EVERY launch of the stream (lead-in and tail included): what a rocprofv3 --stats average of this
      # command shows -- in the lead-in the pairs are admitted at the GPU's own rate and every GEMM has the previous pair's
      # chain DP, verify and sort beside it
      all_ms = [tm["device"]["gemm_ms"] for tm in tms if tm.get("device", {}).get("gemm_ms")]
      all_flops = [tm["device"]["gemm_flops"] for tm in tms if tm.get("device", {}).get("gemm_ms")]
      gemm_tf_all = sum(all_flops) / (sum(all_ms) * 1e-3) / 1e12 if all_ms else 0.0
      res = {
        "metric": "aligned audio-hours/sec", "value": value, "unit": "audio-hours/s", "n_gpus": world,
        "ranks_in_group": grp.group_size(),
        "steps": steps, "warmup": declared_warmup, "ms_per_step": 1e3 * elapsed / steps,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if prec_name == "f32" else "bf16", "data": "synthetic",
        "config": {"workload": wl["desc"] + f", {prec_name} similarity GEMM", "pairs_per_rank_per_step": 1,
                   "video_seconds": wl["seconds"], "audio_seconds": round(pair.audio_seconds, 1),
                   "channels": wl["channels"], "parallelism": f"pairs sharded over {world} GPU(s), no collectives",
                   "distinct_pairs_in_rotation": distinct, "seeds": [5 + rank + j for j in range(distinct)]},
        "realtime_factor": wl["seconds"] * world * steps / elapsed,
        "lead_in_pairs_actual": warmup, "warmup_effective": warmup,
        "whole_stream_value": hours * world * total / (t_end - t_start),
        "timed_region": {"kind": "steady state of one primed pipeline" if pipe is not None else "sequential align() calls",
                         "pairs_streamed": total, "untimed_lead_in_pairs": warmup, "tail_pairs": tail,
                         "whole_stream_s": round(t_end - t_start, 2), "timed_s": round(elapsed, 3),
                         "note": "`warmup` echoes the command line; the untimed lead-in actually streamed is `lead_in_pairs_actual` "
                                 "(at least two LP solves per host worker), `whole_stream_value` is the rate over every pair streamed, "
                                 "lead-in and tail included"},
        "single_pair_latency_s": round(acc["align_s"] / k, 3),
        "single_pair_realtime_factor": round(wl["seconds"] / (acc["align_s"] / k), 1),
        "roofline": {"bound": "mfma", "kernel": "k_match_" + prec_name, "achieved": gemm_tf, "peak": peak, "unit": "TFLOP/s",
                     "frac": gemm_tf / peak, "traffic": None,
                     "frac_all_launches": gemm_tf_all / peak, "avg_launch_ms_all_launches": sum(all_ms) / max(1, len(all_ms)),
                     "launches_all": len(all_ms),
                     "avg_launch_ms": acc["gemm_ms"] / k, "flops_per_launch": acc["gemm_flops"] / k,
                     "algorithmic": "246 flop x (non-quiet audio frames) x (every 4th non-quiet video frame)",
                     "note": "frac / avg_launch_ms = HIP events around the GEMM launches of the TIMED pairs; frac_all_launches / avg_launch_ms_all_launches = "
                             "the same over every launch of the stream, lead-in and tail included: the figure a rocprofv3 --stats average of this command "
                             "supports (profiles/r05_bench_cfg2_bf16_kernel_stats.csv). "
                             "bf16: the kernel is bound by board power, not by its schedule -- a loop of nothing but its MFMAs on random operands holds "
                             "1.86-1.90 GHz on 256 CUs (2.38 on one) = 0.60-0.67 of the 2.4 GHz dense peak, the shipped kernel runs at 1.81-1.84 GHz in-kernel "
                             "with its MFMAs 77 % of the wave cycles (profiles/r05_issue_microbench.txt, r05_pmc_mfma_bf16.json, DESIGN.md 4.3); the devices "
                             "of a pool differ by ~10 % in this kernel's time (alone_launch_ms_same_device is the reference for THIS device)"},
        "feature_stage": {"bound": "hbm", "achieved": acc["feat_bytes"] / (acc["feat_ms"] * 1e-3) / 1e9 if acc.get("feat_ms") else 0.0,
                          "peak": HBM_PEAK_GBS, "unit": "GB/s", "bytes_per_step": acc["feat_bytes"] / k,
                          "ms_per_step": acc["feat_ms"] / k},
        "stage_ms_per_step": {n: round(acc[n] / k, 3) for n in ("feat_ms", "prep_ms", "gemm_ms", "verify_ms", "verify_kernel_ms", "chain_ms",
                                                                "refine_kernel_ms", "refine_dp_ms")},
        "stage_ms_note": "verify_ms = everything between the GEMM and the resident sorted match list (k_verify, radix sort, unpack, row / frame counts, "
                         "with whatever shares the device at that moment); verify_kernel_ms = k_verify alone",
        "host_s_per_step": {"lp": round(acc["lp_s"] / k, 4), "align_latency_per_pair": round(acc["align_s"] / k, 4),
                            "gpu_match_stage_wall": round(acc["match_s"] / k, 4),
                            "chain_enqueue_to_collected": round(acc["chain_s"] / k, 4),
                            "worker_busy": round(acc.get("worker_s", 0.0) / k, 4), "pass1": round(acc.get("pass1_host_s", 0.0) / k, 4),
                            "cluster": round(acc.get("cluster_s", 0.0) / k, 4), "refine": round(acc.get("refine_s", 0.0) / k, 4),
                            "nodes": round(acc.get("nodes_s", 0.0) / k, 4),
                            "gpu_thread": {"pace_wait": round(acc.get("pace_s", 0.0) / k, 4), "features_incl_download": round(acc.get("features_s", 0.0) / k, 4),
                                           "match_begin_to_finish": round(acc["match_s"] / k, 4), "chain_begin": round(acc.get("chain_begin_s", 0.0) / k, 4),
                                           "match_begin_call": round(acc.get("match_begin_s", 0.0) / k, 4),
                                           "collect_chains_under_gemm": round(acc.get("collect_under_gemm_s", 0.0) / k, 4),
                                           "match_finish_call": round(acc.get("match_finish_s", 0.0) / k, 4)},
                            "intervals": {n[3:]: round(acc[n] / k, 4) for n in sorted(acc) if n.startswith("iv_")}},
        "pipeline": {"lp_worker_processes": workers, "gpu_streams": len(gpu_ctxs), "host_cores": os.cpu_count(),
                     "worker_count_rationale": "a quarter more worker processes than this rank's share of the container's CPU quota (align.default_worker_count; "
                                               "without a quota: three per four physical cores): the host stage is CPU-time bound -- the GPU box's container "
                                               "runs under cpu.max = 16 CPUs, which is why every earlier worker sweep was flat beyond 16 "
                                               "(profiles/r06_host_cpu_quota_probe.txt, r06_lp_host_scaling.jsonl)",
                     "note": "GPU stages of pair k+1 and the device chain DPs of earlier pairs overlap the host LP of pair k; "
                             "results identical to sequential align()"},
        "counts": {"gemm_pairs": acc["gemm_pairs"] / k, "survivors": acc["survivors"] / k, "matches": acc["matches"] / k,
                   "path_points": acc["n_path1"] / k, "lp_fit_points": acc["n_fit_points"] / k},
        "max_offset_err_vs_injected_ms": round(inj_err_ms, 3),
        "pcm_resident_in_hbm": not include_h2d,
        "pcm_h2d_ms_audio_side": round(h2d_ms, 2),
        "pcm_h2d_ms_per_step": round(sum(h2d_acc) / max(1, len(h2d_acc)), 2) if include_h2d else None,
      }
      # Which stage limits the stream: the GPU stage (one thread feeds features + prep + GEMM + verify of pair
      # k+1 while the chain DP of pair k runs on its own stream) or the host LP (scipy.optimize.linprog,
      # host-side by contract, one solve per worker process at a time).
      gpu_stage_ms = sum(acc[n] for n in ("feat_ms", "prep_ms", "gemm_ms", "verify_ms")) / k
      gpu_rate = 1e3 / gpu_stage_ms if gpu_stage_ms > 0 else float("inf")
      gpu_rate_wall = k / acc["match_s"] if acc.get("match_s") else float("inf")
      lp_rate = workers * k / acc["lp_s"] if workers > 0 and acc.get("lp_s") else (k / acc["lp_s"] if acc.get("lp_s") else float("inf"))
      res["gpu_stage_pairs_per_s"] = round(min(gpu_rate, gpu_rate_wall), 3)        # per GPU (rank 0's figures)
      res["lp_solves_per_s_rank"] = round(lp_rate, 3)                               # this rank's worker processes
      res["lp_solves_per_s_host"] = round(lp_rate * world, 3)                       # all ranks of the node share the host: rank 0's rate x ranks
      res["measured_pairs_per_s"] = round(world * steps / elapsed, 3)
      methods = {}
      for tm in sel:
        m = str(tm.get("lp_method", "?")).split(" (")[0]
        methods[m] = methods.get(m, 0) + 1
      res["lp_method"] = {"timed_pairs": methods,
                          "note": "tree = the reference's LP solved by HiGHS from the bases of sub-LPs (describealign_amd/lp_tree.py), accepted only with "
                                  "its optimality certificate for the LP as posed; reference = scipy.optimize.linprog(method='highs-ds') from the slack basis"}
      if workers > 0 and acc.get("worker_s"):
        # share of the timed region this rank's worker processes spent inside pass 1 + LP + clustering
        res["lp_worker_utilisation"] = round((steps / elapsed) * (acc["worker_s"] / k) / workers, 3)
      res["bound"] = "host_lp" if lp_rate < 0.9 * min(gpu_rate, gpu_rate_wall) else "gpu"
      # The host stage is bound by CPU TIME: under a cgroup quota (the GPU box: cpu.max = 16 CPUs of a 2 x 64-core host,
      # profiles/r06_host_cpu_quota_probe.txt) it delivers quota / (CPU-seconds per pair) pairs a second, whatever the worker count
      quota = A.cpu_quota()
      res["host_cpu_budget"] = {
          "cgroup_quota_cpus": quota, "logical_cpus": os.cpu_count(),
          "worker_busy_s_per_pair": round(acc.get("worker_s", 0.0) / k, 3),
          "cpu_s_per_pair": {"worker_process": round(acc.get("worker_cpu_s", 0.0) / k, 4), "gpu_feeding_thread": round(acc.get("gpu_thread_cpu_s", 0.0) / k, 4),
                             "refine_thread": round(acc.get("refine_cpu_s", 0.0) / k, 4), "hand_off_thread": round(acc.get("handoff_cpu_s", 0.0) / k, 4),
                             "sum": round(sum(acc.get(n, 0.0) for n in ("worker_cpu_s", "gpu_thread_cpu_s", "refine_cpu_s", "handoff_cpu_s")) / k, 4)},
          "capacity_pairs_per_s": (round(quota / max(1e-9, sum(acc.get(n, 0.0) for n in ("worker_cpu_s", "gpu_thread_cpu_s", "refine_cpu_s", "handoff_cpu_s")) / k), 2)
                                   if quota else None),
          "note": "cgroup cpu.max of this container (None: no quota).  worker_busy_s_per_pair is WALL time inside a worker process (with more "
                  "workers than quota CPUs it includes time spent throttled); cpu_s_per_pair is CPU time (process_time / thread_time) of the pair's "
                  "host work by where it runs (the runtime's helper threads and the interpreter's main thread are not in it); "
                  "capacity_pairs_per_s = quota / that sum: what the host stage can deliver"}
      res["bound_note"] = (f"GPU stage {gpu_stage_ms:.1f} ms of kernels per pair ({gpu_rate:.2f} pairs/s; {gpu_rate_wall:.2f} pairs/s by the feeding "
                           f"thread's wall clock); host LP {acc['lp_s'] / k:.2f} s per solve x {max(1, workers)} worker processes = {lp_rate:.2f} solves/s on "
                           f"this rank's share of the host ({os.cpu_count()} logical CPUs, cgroup CPU quota {A.cpu_quota()}, {world} rank(s): the host stage is "
                           f"CPU-time bound, its capacity is the quota / CPU-seconds per pair)")
      # HBM traffic of the dominant kernel from the committed PMC profile of this workload (PMC
      # collection needs its own rocprofv3 passes; bench.py itself only times with HIP events)
      for prof_name in ("r06_pmc_traffic.json", "r05_pmc_traffic.json", "r04_pmc_traffic.json", "r03_pmc_traffic.json", "r02_pmc_traffic.json", "r01_pmc_traffic.json"):
        try:
          prof = json.load(open(os.path.join(ROOT, "profiles", prof_name)))
          key = workload + "_" + prec_name
          if key in prof:
            res["roofline"]["traffic"] = prof[key]["k_match_" + prec_name]["traffic_bytes_per_launch"]
            res["roofline"]["traffic_source"] = f"profiles/{prof_name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes)"
            # the GEMM's algorithmic output is 12 B per surviving (verified) pair; operand bytes are
            # negligible by the algorithm.  What the kernel moves beyond that (survivor records of the prefilter, the
            # explicit operand streams as far as they miss L2) is this ratio -- its design, not the algorithm.
            alg = 12.0 * acc["matches"] / k
            res["roofline"]["algorithmic_bytes_per_launch"] = alg
            res["roofline"]["traffic_over_algorithmic"] = round(res["roofline"]["traffic"] / alg, 2) if alg > 0 else None
            # the two parts separately: what the prefilter writes (8 B per survivor record, measured in THIS run) and the rest
            # (the operand streams where they miss L2: the resident side re-read once per stripe, the streamed side once per XCD)
            surv_bytes = 8.0 * acc["survivors"] / k
            res["roofline"]["traffic_parts"] = {
                "survivor_records_bytes": surv_bytes, "survivor_records_over_algorithmic": round(surv_bytes / alg, 2) if alg > 0 else None,
                "survivor_records_per_verified_match": round(acc["survivors"] / acc["matches"], 2) if acc.get("matches") else None,
                "operand_streams_bytes": max(0.0, res["roofline"]["traffic"] - surv_bytes),
                "operand_streams_over_algorithmic": round(max(0.0, res["roofline"]["traffic"] - surv_bytes) / alg, 2) if alg > 0 else None}
            res["roofline"]["traffic_note"] = (
                "requests of the L2s to the fabric (mostly served by the 256 MB MALL), per launch: the survivor records of the prefilter "
                "(8 B each, 11 per verified match) and the two explicit operand streams where they miss an XCD's 4 MB L2 (the resident "
                "side once per stripe, the streamed side once per XCD and stripe); traffic / launch time = ~0.2-0.3 TB/s of 8 TB/s -- the kernel is "
                "bound by the matrix cores and board power; voting inside the GEMM (11 -> 1.2 records per match) was measured and moves the "
                "work without saving any (profiles/r05_vote_in_gemm.txt)")
            break
        except Exception:
          pass
      if workers > 0 and out is not None and not quick:
        # the host LP of this pair ONCE MORE with the host otherwise idle (the pipeline has drained): what a solve costs alone
        # against what it cost beside the other workers' solves in the timed region
        try:
          with quiet:
            vf1 = ctx.features_resident(_native.SIDE_VIDEO); af1 = ctx.features_resident(_native.SIDE_AUDIO)
            tm1 = {}
            A.align(vf1, af1, vf1[0], af1[0], ctx=ctx, timings=tm1)
          if tm1.get("device", {}).get("gemm_ms"):
            res["roofline"]["alone_launch_ms_same_device"] = round(tm1["device"]["gemm_ms"], 3)
            res["roofline"]["alone_frac_same_device"] = tm1["device"]["gemm_flops"] / (tm1["device"]["gemm_ms"] * 1e-3) / 1e12 / peak
          res["host_lp"] = {"solve_s_alone": round(tm1["lp_s"], 3), "solve_s_under_load": round(acc["lp_s"] / k, 3), "workers": workers,
                            "solves_per_s_one_worker_alone": round(1.0 / tm1["lp_s"], 3), "solves_per_s_under_load_all_workers": round(lp_rate, 3),
                            "fit_points": int(tm1["n_fit_points"]), "sequential_pair_s_idle_host": round(tm1["total_s"], 3),
                            "helper_processes_of_the_lone_solve": int(tm1.get("lp_helper_processes", 0)),
                            "note": "one sequential align() of this rank's pair after the pipeline has drained (host and GPU otherwise idle): its "
                                    "LP solve beside the mean solve time of the timed pairs inside the worker pool (more workers than quota CPUs: part of "
                                    "the difference is time spent throttled).  A pair aligned on its own spreads the tree's sub-LPs over helper "
                                    "processes (helper_processes_of_the_lone_solve); inside the pool every solve is one process"}
        except Exception as e:
          res["host_lp"] = {"error": str(e)}
      if with_stretch:
        # outside the timed region: the --stretch_audio stage once on this
        # rank's resident pair with the nodes just found
        try:
          t_s = time.perf_counter()
          track, _ = ctx.stretch_resident(out[0], out[1], False)
          wall_s = time.perf_counter() - t_s
          ss = ctx.stats()
          res["stretch_audio_stage"] = {
            "wall_ms_incl_copy_out": round(1e3 * wall_s, 2), "frames": int(track.shape[0]), "channels": int(track.shape[1]),
            "prepare_ms": round(ss["stretch_prepare_ms"], 3), "resample_ms": round(ss["resample_ms"], 3),
            "resample_GBs": round(ss["resample_bytes"] / max(ss["resample_ms"], 1e-9) / 1e6, 1) if ss["resample_ms"] else None,
            "correlate_ms": round(ss["correlate_ms"], 3), "viterbi_ms": round(ss["viterbi_ms"], 3),
            "splice_ms": round(ss["splice_ms"], 3), "finish_ms": round(ss["stretch_finish_ms"], 3),
            "bound": "hbm", "peak_GBs": HBM_PEAK_GBS}
          del track
        except Exception as e:            # never let the auxiliary measurement cost the headline line
          res["stretch_audio_stage"] = {"error": str(e)}
      if with_cpu_baseline:
        spair, (ox, oy), cb = cpu_baseline(wl["channels"], wl["seconds"], args.cpu_sample_seconds, wl["n_jumps"], wl["first_gap"],
                                           golden_case={"cfg2": "e7200s", "cfg1": "e1320"}.get(workload) if rank == 0 else None)
        # same sample through the GPU path: max |node time| difference vs the CPU reference port
        with quiet:
          vf = ctx.features(spair.video, _native.SIDE_VIDEO); af = ctx.features(spair.audio, _native.SIDE_AUDIO)
          gx, gy, *_ = A.align(vf, af, vf[0], af[0], ctx=ctx)
        err = float("nan")
        if len(gx) == len(ox):
          err = 1e3 * max(np.max(np.abs(gx - ox)), np.max(np.abs(gy - oy)))
        cb["max_offset_err_vs_cpu_ms"] = round(err, 4)
        res["cpu_baseline"] = cb
    if pipe is not None:
      pipe.__exit__()
    for sides in rot:
      for st in sides:
        st.close()
    for c in gpu_ctxs[1:]:
      c.close()
    ctx.close()
    return res


def _offset_error_ms(meta, x, y):
  """Every recovered segment's (audio - video) offset against the injected truth at its middle (ms)."""
  err = 0.0
  for k in range(0, len(x) - 1, 2):
    want = meta.true_offset_at(0.5 * (y[k] + y[k + 1]))
    err = max(err, abs((x[k] - y[k]) - want), abs((x[k + 1] - y[k + 1]) - want))
  return 1e3 * err


def finite_batch(bench, workload, seeds):
  """What a STATED batch gets: `seeds` distinct pairs of the workload's shape, their PCM resident in HBM, COLD through one fresh
  pipeline -- no lead-in, no tail; the clock runs from the first submission to the last result (rank-local, then max over ranks).
  With N ranks the seeds are dealt round-robin (describealign.py:1077's directory batch, sharded): configs[3] is seeds 0..31."""
  import contextlib, io
  from describealign_amd import _native, synth
  from describealign_amd import align as A
  args, grp = bench.args, bench.grp
  wl = WORKLOADS[workload]
  prec_name = args.precision or wl["precision"]
  prec = _native.PREC_F32 if prec_name == "f32" else _native.PREC_BF16
  workers = args.pipeline if args.pipeline > 0 else A.default_worker_count(int(os.environ.get("LOCAL_WORLD_SIZE", grp.world)))
  mine = list(seeds)[grp.rank::grp.world]
  ctx = None
  streams, metas = [], []
  t_gen = time.perf_counter()
  # every collective below (the two synchronisations, the reductions) is reached by every rank whatever fails on one of them:
  # a rank that cannot build its share says so (all_ok) instead of leaving the others in a barrier
  def build(sd):
    pair = synth.make_pair(sd, wl["seconds"], n_jumps=wl["n_jumps"], first_gap=wl["first_gap"], channels=wl["channels"])
    sides = []
    for pcm in (pair.video, pair.audio):
      st = _native.PcmStream(bench.device, wl["channels"], pcm.shape[1])
      frames = np.ascontiguousarray(pcm.T)                    # interleaved (n, C), as a decoder delivers it
      for at in range(0, len(frames), 1 << 24):
        st.piece(frames[at:at + (1 << 24)])
      st.sync()
      sides.append(st)
    streams.append(sides)
    metas.append(synth.SynthPair(video=np.empty((wl["channels"], 0), np.int16), audio=np.empty((wl["channels"], 0), np.int16),
                                 jump_video_times=pair.jump_video_times, jump_lengths=pair.jump_lengths, seed=sd))

  setup_err, pipe, t_p = None, None, 0.0
  try:
    if os.environ.get("DALIGN_BENCH_FAIL_FINITE_SETUP") == str(grp.rank):          # test hook: this rank's share cannot be built
      raise RuntimeError("forced by DALIGN_BENCH_FAIL_FINITE_SETUP")
    ctx = _native.Context(bench.device, prec)
    for sd in mine:
      build(sd)
    t_p = time.perf_counter()
    pipe = A.AlignPipeline([ctx], lp_workers=workers)
    pipe.warm()
    t_p = time.perf_counter() - t_p
  except Exception as e:                         # noqa: BLE001
    setup_err = e
  t_gen = time.perf_counter() - t_gen

  def cleanup():
    if pipe is not None:
      try:
        pipe.__exit__()
      except Exception:
        pass
    for sides in streams:
      for st in sides:
        st.close()
    if ctx is not None:
      ctx.close()

  if not grp.all_ok(setup_err is None):
    cleanup()
    raise RuntimeError(f"finite batch: setup failed on {'this' if setup_err is not None else 'another'} rank" + (f": {setup_err}" if setup_err is not None else ""))

  def make_job(k):
    def job(c):
      c.pcm_adopt(_native.SIDE_VIDEO, streams[k][0])          # no copy: the context takes the device buffer over
      c.pcm_adopt(_native.SIDE_AUDIO, streams[k][1])
      return A.RESIDENT_PCM
    return job

  tms, errs, lens = [], [], []
  bench.sync()
  t0 = time.perf_counter()
  run_err = None
  try:
    with contextlib.redirect_stdout(io.StringIO()):
      for k, out in enumerate(pipe.run((make_job(k) for k in range(len(mine))), timings=tms)):
        errs.append(_offset_error_ms(metas[k], out[0], out[1])); lens.append(len(out[0]))
  except Exception as e:                         # noqa: BLE001
    run_err = e
  t1 = time.perf_counter()
  bench.sync()
  cleanup()
  elapsed = grp.max_over_ranks(t1 - t0)
  worst = grp.max_over_ranks(max(errs) if errs else 0.0)
  if not grp.all_ok(run_err is None):
    raise RuntimeError(f"finite batch: failed on {'this' if run_err is not None else 'another'} rank" + (f": {run_err}" if run_err is not None else ""))
  n_all = len(list(seeds))
  if grp.rank != 0:
    return None
  hours = wl["seconds"] / 3600.0
  return {"workload": wl["desc"] + f", {prec_name} similarity GEMM", "pairs": n_all, "seeds": [int(min(seeds)), int(max(seeds))], "distinct_pairs": True,
          "pairs_this_rank": len(mine), "ranks": grp.world, "wall_s": round(elapsed, 3), "value": hours * n_all / elapsed, "unit": "audio-hours/s",
          "realtime_factor": wl["seconds"] * n_all / elapsed, "pairs_per_s": round(n_all / elapsed, 3),
          "lead_in_pairs": 0, "tail_pairs": 0, "lp_worker_processes": workers,
          "mean_lp_s": round(float(np.mean([tm["lp_s"] for tm in tms])), 3) if tms else None,
          "lp_s_per_pair": [round(float(tm["lp_s"]), 3) for tm in tms],          # distinct pairs: how the host LP's time varies with the content
          "lp_fit_points_per_pair": [int(tm.get("n_fit_points", 0)) for tm in tms],
          "lp_method": sorted(set(str(tm.get("lp_method", "?")).split(" (")[0] for tm in tms)),
          "mean_gpu_stage_s": round(float(np.mean([tm["match_s"] for tm in tms])), 4) if tms else None,
          "max_offset_err_vs_injected_ms": round(worst, 3),
          "nodes_per_pair": sorted(set(lens)),
          "untimed": {"synthesis_and_upload_s": round(t_gen - t_p, 1), "pipeline_start_s": round(t_p, 2)},
          "timed_region": "first pair submitted to a fresh, started pipeline -> last result delivered; PCM of every pair resident in HBM "
                          "(PcmStream buffers adopted by the context, no copy); nothing primed, nothing discarded"}


class _FileLock:
  """Ranks that SHARE one device (gloo launch tests) take turns in the matching stage of a tiled pair."""

  def __init__(self, path):
    self.path = path

  def __enter__(self):
    import fcntl
    self.f = open(self.path, "w")
    fcntl.flock(self.f, fcntl.LOCK_EX)

  def __exit__(self, *a):
    import fcntl
    fcntl.flock(self.f, fcntl.LOCK_UN)
    self.f.close()


def tiled_long_pair(bench):
  """BASELINE configs[4]'s path -- the ONE place the alignment path has an exchange step: a single long pair, its matching stage
  tiled over the ranks by audio rows (align.align_tiled), the verified match lists gathered on rank 0 over RCCL / xGMI
  (distrib.Group._gather_device), chain DP + LP + pass 2 on rank 0, result broadcast.  Duration min(8 h, 1 h per rank)."""
  import contextlib, io
  from describealign_amd import _native, synth
  from describealign_amd import align as A
  grp = bench.grp
  secs = float(os.environ.get("DALIGN_BENCH_TILED_SECONDS", min(28800.0, 3600.0 * grp.world)))
  ctx = _native.Context(bench.device, _native.PREC_BF16)
  t0 = time.perf_counter()
  # rank 0 alone synthesises the pair and runs the feature kernel; every rank gets the ten feature rows (120 MB a side at 8 h):
  # eight ranks synthesising 5 GB of PCM each, on a host whose CPU time they share, would take minutes for nothing
  rows, meta, setup_err = None, None, None
  if grp.rank == 0:
    try:
      full = synth.make_pair(13, secs, n_jumps=max(2, int(round(secs / 720.0))), first_gap=min(300.0, secs / 6.0), channels=1)
      vf = ctx.features(full.video, _native.SIDE_VIDEO); af = ctx.features(full.audio, _native.SIDE_AUDIO)
      rows = [np.array(r) for r in list(vf) + list(af)]
      meta = (full.jump_video_times, full.jump_lengths)
      del full, vf, af
    except Exception as e:                       # noqa: BLE001 -- the other ranks are about to wait for the broadcast
      setup_err = e
  if not grp.all_ok(setup_err is None):
    ctx.close()
    raise RuntimeError("tiled long pair: rank 0 could not build the pair" + (f": {setup_err}" if setup_err is not None else ""))
  rows, meta = grp.broadcast_rows(rows, meta)
  vf, af = rows[:5], rows[5:]
  pair = synth.SynthPair(video=np.empty((1, 0), np.int16), audio=np.empty((1, 0), np.int16), jump_video_times=meta[0], jump_lengths=meta[1], seed=13)
  t_gen = time.perf_counter() - t0
  lock = None
  if grp.backend != "nccl" and grp.world > 1:
    lock = _FileLock(os.path.join(os.environ.get("TMPDIR", "/tmp"), f"dalign_bench_tiled_{os.environ.get('MASTER_PORT', '0')}.lock"))
  tm = {}
  bench.sync()
  t1 = time.perf_counter()
  with contextlib.redirect_stdout(io.StringIO()):
    x, y, sim, path, med = A.align_tiled(vf, af, vf[0], af[0], grp, ctx=ctx, timings=tm, match_lock=lock)
  bench.sync()
  el = grp.max_over_ranks(time.perf_counter() - t1)
  match_s = grp.max_over_ranks(tm.get("match_s", 0.0))
  n_local = tm.get("device", {}).get("matches", 0.0)
  ctx.close()
  if grp.rank != 0:
    return None
  total = float(tm["n_matches"])
  moved = 16.0 * max(0.0, total - n_local)                 # packed key + float64 quality per match, rank 0's own block stays put
  return {"workload": f"configs[4] path: ONE synthetic {secs:.0f} s mono pair, {len(pair.jump_lengths)} injected offsets, matching tiled over "
                      f"{grp.world} rank(s) by audio rows, bf16 similarity GEMM", "video_seconds": secs, "ranks": grp.world, "backend": grp.backend,
          "value": (secs / 3600.0) / el, "unit": "audio-hours/s", "realtime_factor": secs / el, "align_s": round(el, 2),
          "match_s_max_over_ranks": round(match_s, 3), "gather_s": round(tm.get("gather_s", 0.0), 3), "chain_s": round(tm.get("chain_s", 0.0), 3),
          "lp_s": round(tm.get("lp_s", 0.0), 2), "refine_s": round(tm.get("refine_s", 0.0), 3), "fit_points": int(tm.get("n_fit_points", 0)),
          "matches": int(total), "gathered_bytes": moved, "gather_GBps": round(moved / max(tm.get("gather_s", 0.0), 1e-9) / 1e9, 2),
          "nodes": int(len(x)), "segments_expected": len(pair.jump_lengths),
          "max_offset_err_vs_injected_ms": round(_offset_error_ms(pair, x, y), 3), "similarity": round(float(sim), 2),
          "untimed": {"synthesis_features_and_broadcast_s": round(t_gen, 1)},
          "note": "the exchange: all-gather of the per-rank match counts, then exact-size point-to-point transfers that land in rank 0's "
                  "context (RCCL: device to device over xGMI; gloo, launch tests only: staged through the host).  The LP of one long pair is ONE "
                  "scipy.optimize.linprog solve on one core (time ~ fit_points^1.8): it, not the GPUs, sets this figure"}


def launch_ranks(n):
  """`python bench.py --gpus N` without a launcher: start `python -m torch.distributed.run --nproc-per-node N bench.py <same
  arguments>` as a child process -- one rank per GPU over RCCL -- BEFORE this process has imported the extension or touched a
  GPU (a process that has initialised the GPU must never be replaced by another program on this pool, so nothing is exec'ed),
  relay rank 0's JSON line and return the child's exit code.  With the RCCL backend fewer than N visible devices is an error,
  never a silent run on fewer GPUs."""
  import socket
  import subprocess
  backend = os.environ.get("DALIGN_DIST_BACKEND", "nccl")
  if backend == "nccl":
    import torch
    have = torch.cuda.device_count()          # counts devices without initialising the runtime
    if have < n:
      print(f"bench.py: --gpus {n} needs {n} visible GPUs for one RCCL rank each, this host shows {have} "
            "(DALIGN_DIST_BACKEND=gloo DALIGN_BENCH_DEVICE=0 runs several ranks on one GPU: launch-path tests only)", file=sys.stderr)
      return 2
  port = os.environ.get("MASTER_PORT")
  if not port:
    with socket.socket() as sk:
      sk.bind(("127.0.0.1", 0))
      port = str(sk.getsockname()[1])
  cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
         "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
  child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, cwd=ROOT)
  lines = []
  for line in child.stdout:                   # the ranks' stderr goes straight through; stdout is relayed line by line
    if line.startswith("{"):
      lines.append(line)
    else:
      sys.stderr.write(line)
  code = child.wait()
  for line in lines[-1:]:
    sys.stdout.write(line)
  sys.stdout.flush()
  if code == 0 and not lines:
    print("bench.py: the ranks exited without printing a result line", file=sys.stderr)
    return 1
  return code


SECONDARY_KEYS = ("value", "unit", "steps", "warmup", "lead_in_pairs_actual", "whole_stream_value", "ms_per_step", "dtype", "config",
                  "realtime_factor", "roofline", "feature_stage", "stage_ms_per_step", "host_s_per_step", "counts", "bound",
                  "gpu_stage_pairs_per_s", "lp_solves_per_s_host", "measured_pairs_per_s", "single_pair_latency_s",
                  "max_offset_err_vs_injected_ms", "timed_region", "host_lp", "lp_worker_utilisation", "lp_method", "cpu_baseline", "warmup_effective", "host_cpu_budget")


def main():
  ap = argparse.ArgumentParser()
  ap.add_argument("--gpus", type=int, default=1)
  ap.add_argument("--steps", type=int, default=24, help="pairs in the timed region")
  ap.add_argument("--warmup", type=int, default=6, help="pairs streamed through the pipeline before the clock starts")
  ap.add_argument("--prime", type=int, default=-1,
                  help="minimum untimed lead-in pairs before the clock starts (-1: two LP solves per worker process; 0: just --warmup)")
  ap.add_argument("--tail", type=int, default=-1, help="pairs submitted behind the timed ones (-1: enough to keep the host stage full)")
  ap.add_argument("--workload", default="cfg2", choices=sorted(WORKLOADS))
  ap.add_argument("--precision", default=None, choices=["f32", "bf16"])
  ap.add_argument("--distinct", type=int, default=4,
                  help="distinct pairs (this rank's seed and the next ones) the stream rotates through, all resident in HBM; 1: the same pair every step")
  ap.add_argument("--no-cpu-baseline", action="store_true")
  ap.add_argument("--cpu-sample-seconds", type=float, default=None)
  ap.add_argument("--no-pcie", action="store_true", help="skip the short PCIe-inclusive measurement reported as `pcie_inclusive`")
  ap.add_argument("--no-secondary", action="store_true", help="skip the configs[1] / configs[3] measurements reported as `secondary` / `secondary_cfg3`")
  ap.add_argument("--no-cfg3", action="store_true", help="skip only the configs[3] measurement")
  ap.add_argument("--no-finite", action="store_true", help="skip the cold finite-batch measurements (`finite_batch_cfg3`: 32 distinct 30 min pairs, "
                                                            "`finite_batch_cfg2`: 8 distinct 2 h pairs)")
  ap.add_argument("--no-tiled", action="store_true", help="N > 1 only: skip the single long pair tiled over the ranks (`secondary_tiled`, the RCCL exchange)")
  ap.add_argument("--include-h2d", action="store_true",
                  help="diagnostic: re-upload the PCM over PCIe inside every step (the PCIe-inclusive rate; never the headline value)")
  ap.add_argument("--gpu-streams", type=int, default=1,
                  help="contexts (HIP streams + host threads) feeding the GPU matching stage; pipelined mode only")
  ap.add_argument("--separate-calls", action="store_true",
                  help="diagnostic: the GPU thread issues features / match_begin / match_finish / chain_begin as separate calls (round-4 behaviour) "
                       "instead of the one native call per pair (da_pair_stage)")
  ap.add_argument("--pipeline", type=int, default=-1,
                  help="host worker processes (pass 1, LP); -1 = sized for this rank's share of the host; 0 = strictly sequential align()")
  args = ap.parse_args()

  if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
    # plain `python bench.py --gpus N`: this process becomes the launcher (it never touches a GPU) and the N ranks run
    # under torch.distributed.run as its CHILD; their one JSON line is relayed
    sys.exit(launch_ranks(args.gpus))
  if "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) != args.gpus:
    print(f"bench.py: --gpus {args.gpus} but the launcher started {os.environ['WORLD_SIZE']} rank(s); the line reports the ranks that ran",
          file=sys.stderr)

  import torch          # noqa: F401  -- FIRST: its bundled HIP runtime must be the one libdalign.so binds to
  from describealign_amd import distrib
  # DALIGN_DIST_BACKEND=gloo and DALIGN_BENCH_DEVICE=<id> exist so that the multi-rank launch path
  # can be exercised on a box with fewer GPUs than ranks (RCCL refuses two ranks on one device)
  grp = distrib.Group(os.environ.get("DALIGN_DIST_BACKEND", "nccl"))
  device = grp.local_rank if grp.world > 1 else 0
  if os.environ.get("DALIGN_BENCH_DEVICE"):
    device = int(os.environ["DALIGN_BENCH_DEVICE"])
  b = Bench(args, grp, device)
  single = grp.world == 1
  res = b.run(args.workload, args.steps, args.warmup, with_cpu_baseline=single and not args.no_cpu_baseline, with_stretch=True)
  if single and not args.include_h2d and not args.no_pcie and res is not None:
    # the same workload with the PCM re-uploaded from page-locked host memory in every step: never the headline value
    px = b.run(args.workload, args.steps, args.warmup, with_cpu_baseline=False, with_stretch=False, include_h2d=True)
    res["pcie_inclusive"] = {"value": px["value"], "unit": px["unit"], "steps": px["steps"], "warmup": px["warmup"],
                             "ms_per_step": px["ms_per_step"], "h2d_ms_per_step": px["pcm_h2d_ms_per_step"],
                             "h2d_bytes_per_step": int(2 * sum(a.size for a in (b.pairs[args.workload].video, b.pairs[args.workload].audio))),
                             "note": "PCM in page-locked host memory (da_host_alloc), uploaded asynchronously on the copy stream "
                                     "(da_pcm_upload_async) in every step; value = whole-step rate including that transfer"}
  if single and args.workload == "cfg2" and not args.no_secondary and args.precision is None:
    # timed regions of >= 10 s (cfg1: ~47 ms per pair, cfg3: ~70 ms): a 3-5 s region reads +-10 % from one host to the next
    sec = b.run("cfg1", max(args.steps, 256), max(args.warmup, 8), with_cpu_baseline=not args.no_cpu_baseline, with_stretch=False)
    if res is not None and sec is not None:
      res["secondary"] = {k: sec[k] for k in SECONDARY_KEYS if k in sec}
    if not args.no_cfg3:
      sec = b.run("cfg3", max(args.steps, 192), max(args.warmup, 8), with_cpu_baseline=False, with_stretch=False)
      if res is not None and sec is not None:
        res["secondary_cfg3"] = {k: sec[k] for k in SECONDARY_KEYS if k in sec}
  if args.workload == "cfg2" and not args.no_secondary and args.precision is None and not args.no_finite:
    # what a stated batch gets, cold (every rank takes part: the seeds are dealt over the ranks)
    for key, wlname, seeds in (("finite_batch_cfg3", "cfg3", range(0, 32)), ("finite_batch_cfg2", "cfg2", range(5, 13))):
      try:
        fb = finite_batch(b, wlname, seeds)
      except Exception as e:            # never let an auxiliary measurement cost the headline line
        fb = {"error": f"{type(e).__name__}: {e}"}
      if res is not None:
        res[key] = fb
  if not single and not args.no_tiled:
    try:
      tl = tiled_long_pair(b)
    except Exception as e:
      tl = {"error": f"{type(e).__name__}: {e}"}
    if res is not None:
      res["secondary_tiled"] = tl
  if grp.rank == 0:
    print(json.dumps(res))
  grp.close()


if __name__ == "__main__":
  main()

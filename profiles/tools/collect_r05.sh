#!/bin/bash
# Round-5 evidence, command by command (each line is what was run on a lease box from the repo root; outputs land in
# gpurun_out/r05/ and the summaries kept are the profiles/r05_* files named on the right).
#
#   bash profiles/tools/r05_shape_bench.sh                                   -> profiles/r05_issue_microbench.txt      (MFMA shapes for K = 41)
#   bash profiles/tools/r05_cu_mask_sweep.sh ":" "4:" "8:" "8:rest" "12:" "x1:" "x2:" "16:rest"
#                                                                            -> profiles/r05_cu_mask_sweep.jsonl       (chain DP's CU mask)
#   bash profiles/tools/r05_vote_in_gemm.sh                                  -> profiles/r05_vote_in_gemm.{txt,jsonl}  (vote inside the GEMM: ablation)
#   bash profiles/tools/pmc_match.sh bf16 7200 cfg2 2; ... bf16 1800 cfg3 1; ... f32 1320 cfg1 1
#                                                                            -> profiles/r05_pmc_mfma_bf16.json, r05_pmc_mfma_f32.json, r05_pmc_traffic.json
#   make -C describealign_amd/csrc variant NAME=clock EXTRA=-DDA_DBG_BF_CLOCK; DALIGN_LIB=.../libdalign_clock.so python3 tests/gpu_probe_idle.py cfg2
#                                                                            -> in_kernel_clock_of_the_shipped_kernel in r05_pmc_mfma_bf16.json
#   make -C describealign_amd/csrc fvariant NAME=nofir EXTRA=-DDA_DBG_FEAT_NOFIR; python3 profiles/tools/feature_bench.py (shipped, then DALIGN_LIB=...nofir.so)
#                                                                            -> profiles/r05_features_ablation.txt
#   DALIGN_DEBUG_TIMES=1 python3 bench.py --workload cfg1 --steps 160 ... 2> stamps.txt; python3 profiles/tools/stage_stamps.py stamps.txt
#   python3 tests/gpu_probe_{gaps,stage_loop,idle,pageable,refine_beside}.py   -> profiles/r05_pipeline_stalls.txt       (hipStreamSynchronize's late wake-ups)
#   bash profiles/tools/r05_workers_cfg1.sh cfg1:24 ... cfg3:40              -> profiles/r05_workers_short_pairs.jsonl
#   bash profiles/tools/r05_final.sh                                         -> profiles/r05_bench_cfg{2_bf16,1_f32}_kernel_stats.csv, r05_trace_cfg{1,2}_gaps.json,
#                                                                               r05_bench_default.json, r05_bench_driver_flags.jsonl (python3 bench.py x 3)
#   python3 tests/gpu_stress_chain.py 240; python3 tests/gpu_stress_pipeline.py 48 8; python3 tests/gpu_stress_resident.py
#                                                                            -> 28 996 chain instances / 38 pipeline pairs / 312 resident pairs, 0 mismatches
#
# NOTE on kernel traces: rocprofv3 --kernel-trace serialises dispatches of different queues; kernel DURATIONS in its CSVs are good,
# the gaps between kernels of different streams are not what an unprofiled run does (profiles/r05_pipeline_stalls.txt).
echo "this file is a record; run the lines above one at a time on a GPU box"

#!/bin/bash
# Runs on the GPU box: the measurements of round 5 that are kept under profiles/ (stamped by tools/stamp_profiles.py).
# Usage: tools/r05_measure.sh <part>   part = headline | pmc | round | all
set -u
cd $GRAFT_REPO_ROOT
PART=${1:-all}
if [ $PART = headline ] || [ $PART = all ]; then
  # 1. the headline command under rocprofv3 --kernel-trace --stats and the two HBM-traffic counters (separate passes)
  bash tools/profile_bench.sh r05 --steps 3 --warmup 1 > gpurun_out/r05_headline.log 2>&1
  # ... and the default line itself (other workloads + the library-default workspace figure), untraced
  python3 bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err
  echo "headline done"
fi
if [ $PART = pmc ] || [ $PART = all ]; then
  # 2. counter sets AT SIZE: the C2 workload itself (K = 69 kernel) and C4 (K = 256 wave-group kernel)
  bash tools/prof_counters.sh r05_c2_at_size > gpurun_out/r05_pmc_c2.log 2>&1
  bash tools/prof_counters.sh r05_c4_at_size --workload c4 > gpurun_out/r05_pmc_c4.log 2>&1
  bash tools/prof_counters_cmd.sh r05_identify id_ tools/measure_configs.py identify > gpurun_out/r05_pmc_identify.log 2>&1
  echo "pmc done"
fi
if [ $PART = round ] || [ $PART = all ]; then
  bash tools/measure_round.sh r05 > gpurun_out/r05_round.log 2>&1
  python3 tools/measure_configs.py k320 k350 k402 k448 k500 k520 k600 k640 k700 k768 k1000 k1100 > gpurun_out/round_r05/wide_beyond_256.jsonl 2> gpurun_out/round_r05/wide_beyond_256.err
  echo "round done"
fi
if [ $PART = extra ] || [ $PART = extra_nopmc ] || [ $PART = all ]; then
  mkdir -p gpurun_out/round_r05
  # round 5: the consumers of a small launch with two waves per window and on the one-wave kernels (C1 shape), the C1
  # launch's counters (a wave alone on its SIMD), the product path's timeline, the allocation-cost curve
  python3 tools/measure_configs.py c1_consumers > gpurun_out/round_r05/c1_consumers.jsonl 2> gpurun_out/round_r05/c1_consumers.err
  if [ $PART != extra_nopmc ]; then bash tools/prof_counters.sh r05_c1_lone_wave --workload c1 > gpurun_out/r05_pmc_c1.log 2>&1; fi
  FSMC_HOST_TIMING=1 python3 tools/measure_configs.py run_c2 > gpurun_out/round_r05/run_c2_timeline.json 2> gpurun_out/round_r05/run_c2_timeline.err
  python3 tools/measure_configs.py run_c1_asmc > gpurun_out/round_r05/run_c1_asmc.jsonl 2> gpurun_out/round_r05/run_c1_asmc.err
  python3 tools/malloc_cost.py > gpurun_out/round_r05/malloc_cost.txt 2>&1
  python3 bench.py --workload c1 --mode sums --steps 10 --warmup 2 --cpu-pairs 0 > gpurun_out/round_r05/c1_sums_bench.json 2> gpurun_out/round_r05/c1_sums_bench.err
  echo "extra done"
fi

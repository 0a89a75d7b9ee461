#!/bin/bash
# Regenerates the judged artefacts under profiles/ on the GPU box (run through gpurun from the repo root):
#   ${R}_bench.json                 default bench.py line (no profiler attached)
#   ${R}_bench_kernel_stats.txt     rocprofv3 --kernel-trace --stats summary of the same command + timed-region table
#   ${R}_bench_under_rocprof.json   the bench line printed while the profiler was attached
#   ${R}_timeline.txt               one training step, kernel by kernel
#   ${R}_timeline_grid_update_step.txt   a step that starts with a density-grid update
#   ${R}_pmc_traffic.json           FETCH_SIZE / WRITE_SIZE / TCC_EA0_ATOMIC of the grid kernels and Adam (separate --pmc passes)
# Traces go to /tmp (they exceed the 64 MiB that travels back); only summaries are copied.
# usage: bash tools/collect_profiles.sh [nopmc]      (R=r03 by default)
set -e -o pipefail
R=${R:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/prof
python3 bench.py > gpurun_out/prof/bench.log 2>&1
tail -1 gpurun_out/prof/bench.log > gpurun_out/prof/${R}_bench.json
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_stats -- python3 bench.py --no-cpu-baseline > gpurun_out/prof/bench_rocprof.log 2>&1
grep '^{"metric"' gpurun_out/prof/bench_rocprof.log | tail -1 > gpurun_out/prof/${R}_bench_under_rocprof.json
STATS=$(ls /tmp/p_stats/*/*_kernel_stats.csv | head -1)
TRACE=$(ls /tmp/p_stats/*/*_kernel_trace.csv | head -1)
python3 tools/prof_summary.py "$STATS" > gpurun_out/prof/${R}_bench_kernel_stats.txt
echo >> gpurun_out/prof/${R}_bench_kernel_stats.txt
python3 tools/timeline.py "$TRACE" region 40 8 >> gpurun_out/prof/${R}_bench_kernel_stats.txt
python3 tools/timeline.py "$TRACE" 12 > gpurun_out/prof/${R}_timeline.txt
python3 tools/timeline.py "$TRACE" update 10 > gpurun_out/prof/${R}_timeline_grid_update_step.txt
python3 tools/timeline.py "$TRACE" steps 40 8 > gpurun_out/prof/${R}_step_walls.txt
if [ "$1" != "nopmc" ]; then
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/p_fetch -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 4 > gpurun_out/prof/pmc_fetch.log 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d /tmp/p_write -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 4 > gpurun_out/prof/pmc_write.log 2>&1
  rocprofv3 --pmc TCC_EA0_ATOMIC_sum --output-format csv -d /tmp/p_atomic -- python3 bench.py --no-cpu-baseline --steps 10 --warmup 4 > gpurun_out/prof/pmc_atomic.log 2>&1
  SAMPLES=$(python3 -c "import json;print(json.load(open('gpurun_out/prof/${R}_bench.json'))['kernels']['grid_bwd_param']['avg_samples'])")
  PARAMS=$(python3 -c "import json;print(2*json.load(open('gpurun_out/prof/${R}_bench.json'))['kernels']['adam_step']['avg_params'])")
  python3 tools/pmc_traffic.py /tmp/p_fetch /tmp/p_write 10 "$SAMPLES" /tmp/p_atomic "$PARAMS" > gpurun_out/prof/${R}_pmc_traffic.json
fi
ls -la gpurun_out/prof

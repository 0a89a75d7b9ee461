cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/r02
for v in 0 1 0 1 0 1 0 1; do NGP_NO_NORM_BOUND=$v python bench.py --no-cpu-baseline --steps 80 > gpurun_out/r02/bench_nb$v.log 2>&1; echo "no_bound=$v $(grep -o '"ms_per_step": [0-9.]*' gpurun_out/r02/bench_nb$v.log | head -1)"; done

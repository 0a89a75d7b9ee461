export NGP_AB_VARIANTS=1
python -c "import ngp_amd" > gpurun_out/exp7_build.log 2>&1
python -c "
import torch
print('prio range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else None)
for p in (-2,-1,0,1,2):
    print(p, torch.cuda.Stream(priority=p).priority)
"
run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-solo --windows 3 > gpurun_out/exp7_$name.log 2>&1 || { echo "$name failed"; tail -5 gpurun_out/exp7_$name.log; return 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/exp7_$name.log") if l.startswith("{")][-1])
k=d["kernels"]
print("%-22s ms/step %.3f median %.3f | adam %.3f fwd %.3f dgrad %.3f wgrad %.3f gfwd %.3f gbwdin %.3f scat %.3f"%("$name",d["ms_per_step"],d["ms_per_step_median_of_windows"],k["adam_step"]["avg_ms"],k["mlp2_fwd"]["avg_ms"],k["mlp_bwd_input"]["avg_ms"],k["mlp_bwd_weight"]["avg_ms"],k["grid_fwd"]["avg_ms"],k["grid_bwd_input"]["avg_ms"],k["grid_bwd_param"]["avg_ms"]))
PY
}
for rep in 1 2; do
run new A=1 &&
run main_hi NGP_MAIN_PRIO=-1 &&
run main_hi_fwd_hi NGP_MAIN_PRIO=-1 NGP_FWD_PRIO=-1 &&
run main_hi_opt_hi NGP_MAIN_PRIO=-1 NGP_OPT_PRIO=-1 || exit 1
done

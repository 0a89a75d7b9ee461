run() {  # name, env...
  name=$1; shift
  env "$@" timeout -k 10 200 python bench.py --steps 60 --warmup 10 --no-cpu-baseline --no-solo --windows 3 > gpurun_out/exp9_$name.log 2>&1 || { echo "$name failed"; tail -5 gpurun_out/exp9_$name.log; return 1; }
  python - <<PY
import json
d=json.loads([l for l in open("gpurun_out/exp9_$name.log") if l.startswith("{")][-1])
k=d["kernels"]
print("%-22s ms/step %.3f median %.3f | adam %.3f fwd %.3f dgrad %.3f wgrad %.3f gfwd %.3f gbwdin %.3f scat %.3f"%("$name",d["ms_per_step"],d["ms_per_step_median_of_windows"],k["adam_step"]["avg_ms"],k["mlp2_fwd"]["avg_ms"],k["mlp_bwd_input"]["avg_ms"],k["mlp_bwd_weight"]["avg_ms"],k["grid_fwd"]["avg_ms"],k["grid_bwd_input"]["avg_ms"],k["grid_bwd_param"]["avg_ms"]))
PY
}
for rep in 1 2 3; do
run new A=1 &&
run heads_beside NGP_HEADS_BESIDE=1 || exit 1
done

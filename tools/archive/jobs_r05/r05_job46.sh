#!/bin/bash
# Queue priorities for the side streams of the class kernels (RK_CLASS_PRIO = priorities of R = 2, 3, 4; R = 1 on the caller's stream):
# does a high-priority R = 3 / R = 4 queue pull the kernels that end last forward? With graph replay and with direct launches.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job46
mkdir -p $O
python3 - <<'PY'
import ctypes
h = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
print("hipDeviceGetStreamPriorityRange ->", h.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), "least", lo.value, "greatest", hi.value)
PY
one() {
  python3 bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-pageable-leg > $O/b.json 2> $O/b.err
  python3 -c '
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("%-28s device-resident %.3f ms (kernel %.3f) seam %.3f (kernel %.3f)" % (sys.argv[2], d["ms_per_step_device_resident"], d["kernel_ms_device_resident"], d["ms_per_step"], d["kernel_ms"]))
' $O/b.json "$1" || tail -3 $O/b.err
}
for rep in 1 2; do
for prio in "0,0,0" "1,-1,-1" "0,-1,-1" "0,-1,0" "1,-1,0" "0,0,-1"; do
  RK_CLASS_PRIO=$prio one "graph prio $prio"
  RK_CLASS_PRIO=$prio RK_GRAPH=0 RK_HOST_GRAPH=0 one "direct prio $prio"
done
done 2>&1 | tee $O/out.txt

#!/bin/bash
# The host side split into four translation units (rk_state / rk_launch / rk_host_out / rk_replica): the whole GPU suite twice, smoke,
# the default bench line, the leapfrog harness.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r06_job10
mkdir -p $O
for rep in 1 2; do
  timeout 2400 python3 -m pytest tests -m gpu -x -q 2>&1 | tail -3 | tee -a $O/pytest_gpu.txt
done
timeout 300 python3 -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee $O/smoke.txt
timeout 600 python3 bench.py 2>&1 | tail -1 > $O/bench_4m.json
python3 -c "
import json;d=json.loads(open('$O/bench_4m.json').read());print(d['value'],d['ms_per_step'],d['roofline']['frac'],d.get('value_device_resident'),d.get('value_host_outputs_pageable'))" | tee $O/bench_4m.txt
make -C examples > /dev/null 2>&1
for n in 100000 350000 1000000 2000000 3000000 4000000 8000000; do
  echo -n "$n " | tee -a $O/leapfrog.txt
  timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("nparts %d step %.4f rebuild %.4f traversal %.4f" % (g("nparts") if "nparts" in l else 0, g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
done

#!/bin/bash
# Same-box A/B/C of the rebuild: start of round 5's second session (00c64c1), after the sort / masks / block maxima (HEAD~1 of this job's
# commit), now (level bytes, fused windows, DPP boxes, wide loads in k_pack_nodes). Interleaved, three repetitions.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
O=$ROOT/gpurun_out/r05_job45
mkdir -p $O
make -C examples > /dev/null 2>&1
for rep in 1 2 3; do
  for n in 100000 350000 1000000 2000000 4000000; do
    for v in r5start sortfix now; do
      if [ $v = now ]; then unset LD_LIBRARY_PATH; else export LD_LIBRARY_PATH=$ROOT/rakau_amd/lib_exp_$v; fi
      echo -n "$v $n " | tee -a $O/leapfrog.txt
      timeout 300 examples/leapfrog --nparts $n --steps 40 --warmup 5 2>&1 | tail -1 | python3 -c '
import sys,re
l=sys.stdin.read()
g=lambda k: float(re.search("\"%s\": ([0-9.]+)" % k, l).group(1))
print("step %.4f rebuild %.4f traversal %.4f" % (g("ms_per_step"), g("ms_rebuild"), g("ms_traversal")))' | tee -a $O/leapfrog.txt
    done
  done
done

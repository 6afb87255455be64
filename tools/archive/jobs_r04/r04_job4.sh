#!/bin/bash
# PMC digest of the 4M step with the common lists evaluated by k_common.
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
export RK_COMMON=1
BENCH_ARGS="" bash $ROOT/tools/prof_pmc.sh gpurun_out/r04_job4/pmc > /dev/null 2>&1
python3 $ROOT/tools/pmc_summary.py $ROOT/gpurun_out/r04_job4/pmc > $ROOT/gpurun_out/r04_job4/pmc_summary.txt 2>&1
python3 $ROOT/tools/pmc_digest.py $ROOT/gpurun_out/r04_job4/pmc_summary.txt | tee $ROOT/gpurun_out/r04_job4/pmc_digest.txt
find $ROOT/gpurun_out/r04_job4/pmc -name "*.csv" -size +200k -delete

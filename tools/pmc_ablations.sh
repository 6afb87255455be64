cd /tmp && export TMPDIR=/tmp
for v in ab0 abL abE; do
RAKAU_AMD_LIB=$GRAFT_REPO_ROOT/rakau_amd/lib_$v/librakau_amd.so rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_$v -- python3 $GRAFT_REPO_ROOT/bench.py --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
echo $v; python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $GRAFT_REPO_ROOT/gpurun_out/pmc_$v | grep -A3 "k_list" | grep "VALU\|k_list"
done

#!/bin/bash
# Between 30 000 and 60 000 critical nodes: the class lists read backwards on k_list_any (RK_PLAN_REV_MAX_GROUPS=60000, default)
# against the light-tail plan on the class kernels (=30000: nothing takes the backwards arrangement).
set -u
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
cd $ROOT
S=1150000,1300000,1500000,1800000,2000000,2300000
for rep in 1 2; do
for v in 60000 30000; do
  echo "RK_PLAN_REV_MAX_GROUPS=$v $(RK_PLAN_REV_MAX_GROUPS=$v timeout 900 python3 tools/size_scan.py $S 2>&1 | grep -v amdgpu | tail -1)"
done; done
echo "RK_PLAN_MAX_GROUPS=20000 RK_PLAN_REV_MAX_GROUPS=20000 $(RK_PLAN_MAX_GROUPS=20000 RK_PLAN_REV_MAX_GROUPS=20000 timeout 900 python3 tools/size_scan.py 800000,1000000,1150000 2>&1 | grep -v amdgpu | tail -1)"
echo "default $(timeout 900 python3 tools/size_scan.py 800000,1000000,1150000 2>&1 | grep -v amdgpu | tail -1)"

#!/bin/bash
# A/B of library variants on one kernel variant: tools/ab_split.sh <n> <variant> <name>...   (name "base" = in-tree library)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
N=$1; V=$2; shift; shift
for rep in 1 2; do
for v in "$@"; do
  if [ $v = base ]; then lib=$ROOT/rakau_amd/lib/librakau_amd.so; else lib=$ROOT/rakau_amd/lib_$v/librakau_amd.so; fi
  echo -n "$v: "; RAKAU_AMD_LIB=$lib timeout 200 python3 $ROOT/tools/run_variant.py $N $V 40 2>&1 | tail -1
done; done

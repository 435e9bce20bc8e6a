#!/bin/bash
# A/B of two builds of the library on the Family B 100-qubit step and the level-1 row kernels:
#   bash scripts/ab_lib.sh <lib A> <lib B> [batch]      (paths relative to the repo root; "-" = the shipped library)
R=${GRAFT_REPO_ROOT:-/root/repo}
B=${3:-64}
for lib in "$1" "$2"; do
  if [ "$lib" = "-" ]; then unset MLQEM_LIB; else export MLQEM_LIB=$R/$lib; fi
  echo "== library: $lib"
  python3 $R/scripts/deg_sort_probe.py 20 2>&1 | grep -v amdgpu.ids
  python3 $R/scripts/profile_family_b.py $B 12 100 2>&1 | grep "family B"
done

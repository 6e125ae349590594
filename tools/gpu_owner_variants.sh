#!/bin/bash
# tools/owner_bench.py over library variants (tools/var_<name>.so) on ONE box.  usage: gpu_owner_variants.sh <tag> variant...
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/$1; shift; mkdir -p $O; cd $R
cp $R/difacto_amd/libdifacto_hip.so /tmp/keep.so
python tools/owner_bench.py 300 64 8 entry 2>&1 | tail -1 | cut -c1-300 | tee $O/base_entry.txt
python tools/owner_bench.py 300 64 8 listed 2>&1 | tail -1 | cut -c1-300 | tee $O/base_listed.txt
for v in "$@"; do
  cp $R/tools/var_$v.so $R/difacto_amd/libdifacto_hip.so
  echo "== $v"; python tools/owner_bench.py 300 64 8 listed 2>&1 | tail -1 | cut -c1-300 | tee $O/var_$v.txt
done
cp /tmp/keep.so $R/difacto_amd/libdifacto_hip.so

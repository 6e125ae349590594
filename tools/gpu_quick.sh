#!/bin/bash
# quick GPU check: the given pytest selection (+ host tests with HOSTTESTS=1)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R; mkdir -p gpurun_out/quick
timeout 900 python -m pytest tests -m gpu -x -q "$@" > gpurun_out/quick/pytest.log 2>&1
grep -vE "amdgpu.ids|socket.cpp|Gloo|^$|NCCL|longer_pathname" gpurun_out/quick/pytest.log | tail -60 | cut -c1-400
[ -n "$HOSTTESTS" ] && timeout 300 ./build/difacto_host_tests tests/golden/rcv1_100.libsvm 2>&1 | tail -14

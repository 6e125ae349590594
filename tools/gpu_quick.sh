#!/bin/bash
# quick GPU check: the given pytest selection + host tests
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R; mkdir -p gpurun_out/quick
timeout 900 python -m pytest tests -m gpu -x -q "$@" 2>&1 | grep -vE "amdgpu.ids|socket.cpp|Gloo|^$" | tail -15 | cut -c1-300
timeout 300 ./build/difacto_host_tests tests/golden/rcv1_100.libsvm 2>&1 | tail -14

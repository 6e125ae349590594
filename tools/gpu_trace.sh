#!/bin/bash
# per-block role / span trace of one k_backward_all launch (tools/var_trace.so = the library built with -DDFH_BWD_TRACE)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
cp difacto_amd/libdifacto_hip.so /tmp/keep.so; cp tools/var_trace.so difacto_amd/libdifacto_hip.so
timeout 300 python tools/bwd_trace.py 2>&1 | tail -40
cp /tmp/keep.so difacto_amd/libdifacto_hip.so

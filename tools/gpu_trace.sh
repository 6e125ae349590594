#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
cp difacto_amd/libdifacto_hip.so /tmp/keep.so; cp tools/var_trace.so difacto_amd/libdifacto_hip.so
timeout 300 python tools/bwd_trace.py 2>&1 | grep -v amdgpu.ids
cp /tmp/keep.so difacto_amd/libdifacto_hip.so

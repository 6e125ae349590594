#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
sed 's/V_init = refrand/V_init = hash/; s/max_num_epochs = 10/max_num_epochs = 2/' example/rcv1_fm.conf > /tmp/c.conf
export DMLC_ROLE=worker DMLC_NUM_WORKER=1 DIFACTO_RANK=0 DIFACTO_DEVICE=0 DIFACTO_TRACE=1
timeout -s KILL 30 ./build/difacto argfile=/tmp/c.conf > /tmp/outA.txt 2>&1; echo "rc=$?"; grep -v "NCCL\|^$" /tmp/outA.txt | tail -30 | cut -c1-220

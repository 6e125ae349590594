#!/bin/bash
# Train on N GPUs of this node, one process per GPU, the model row-sharded by key range
# (difacto_amd/host/sharded_store.h).  usage: example/run_local_gpus.sh N key=val ... [argfile=x.conf]
# The data is cut into N x num_jobs_per_epoch parts; add V_init=hash (order-independent V init).
N=${1:?number of GPUs}; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
RV=$(mktemp -u /tmp/difacto_rendezvous.XXXXXX)
pids=()
for ((r = 0; r < N; r++)); do
  DMLC_ROLE=worker DMLC_NUM_WORKER=$N DIFACTO_RANK=$r DIFACTO_RENDEZVOUS=$RV HSA_ENABLE_IPC_MODE_LEGACY=0 \
    "$ROOT/build/difacto" "$@" V_init=hash 2> >(sed "s/^/[rank $r] /" >&2) &
  pids+=($!)
done
rc=0
for p in "${pids[@]}"; do wait $p || rc=$?; done
rm -f "$RV"
exit $rc

#!/bin/bash
# round 4: where the end-to-end loop's time goes WITHOUT the tracer (host sections of the worker loop, of
# dfh_batch_prepare_rows and of the uploads), on the 19.2 M-row .rec file of tools/gpu_r04p.sh; upload threads 1 / 2 / 4
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04u; mkdir -p $O; cd $R
cat > /tmp/mk.py <<'PY'
import os, sys, numpy as np
sys.path.insert(0, os.environ["R"])
from oracle import ingest as oi
rows=400000; rng=np.random.default_rng(1)
ints = rng.zipf(1.3, size=(rows, 13)) % 10000
cats = (rng.zipf(1.1, size=(rows, 26)) % 1000000).astype(np.uint64) * np.uint64(2654435761) % np.uint64(1 << 32)
lab = (rng.random(rows) < 0.25).astype(np.float32)
tok = np.concatenate([ints.astype(np.uint64), cats], 1)
idx = ((tok * np.uint64(0x9E3779B97F4A7C15)) << np.uint64(12) | np.arange(39, dtype=np.uint64)[None, :]).reshape(-1)
recs=[]
for a in range(0, rows, 10000):
    o=(np.arange(10001)*39).astype(np.uint64)
    recs.append(oi.write_crb_record(o, lab[a:a+10000], idx[a*39:(a+10000)*39]))
blob=oi.write_recordio(recs)
with open("/tmp/big.rec","wb") as f:
    for _ in range(48): f.write(blob)
with open("/tmp/small.rec","wb") as f: f.write(blob)
print("written", os.path.getsize("/tmp/big.rec")/1e6, "MB")
PY
R=$R python /tmp/mk.py
ARGS="data_format=rec task=train learner=sgd batch_size=10000 max_num_epochs=1 V_dim=64 V_threshold=0 l1=0 lr=.01 V_lr=.01 V_init=hash table_capacity=8388608 stop_rel_objv=0 num_jobs_per_epoch=1"
run() {  # name env...
  n=$1; shift
  for f in small big big; do
    t0=$(date +%s.%N)
    env "$@" $R/build/difacto data_in=/tmp/$f.rec $ARGS > $O/run_${n}_$f.log 2>&1
    echo "$n $f wall $(python3 -c "import time,sys; print(round(time.time()-float(sys.argv[1]),3))" $t0) s"
  done
  grep -hE "host loop|reader: |prepare_rows x|load_host x|batch reader" $O/run_${n}_big.log | sed 's/.*\] //' | cut -c1-330 | sort | uniq -c | sort -rn | head -12
}
run plain A=1
run prof DIFACTO_PROFILE=1 DFH_PROFILE_PREP=1
run up1 DIFACTO_PROFILE=1 DIFACTO_UPLOAD_THREADS=1
run up4 DIFACTO_PROFILE=1 DIFACTO_UPLOAD_THREADS=4
run spin DIFACTO_PROFILE=1 DFH_SCHEDULE=spin

#!/bin/bash
# kernel trace of build/difacto on the 19.2 M-row .rec file: which kernels the device feed's 0.27 ms per minibatch are made of
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04j; mkdir -p $O; cd $R
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
print("written", os.path.getsize("/tmp/big.rec")/1e6, "MB")
PY
R=$R python /tmp/mk.py
ARGS="data_in=/tmp/big.rec data_format=rec task=train learner=sgd batch_size=10000 max_num_epochs=1 V_dim=64 V_threshold=0 l1=0 lr=.01 V_lr=.01 V_init=hash table_capacity=8388608 stop_rel_objv=0 num_jobs_per_epoch=1"
( time DIFACTO_PROFILE=1 $R/build/difacto $ARGS ) 2>&1 | grep -E "host loop|real|Training" | cut -c1-200
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $O/prof -o kt -- $R/build/difacto $ARGS > $O/prof.log 2>&1
DB=$(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1)
python $R/tools/rocpd_stats.py $DB $O/kernel_stats_e2e_rec.txt > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $DB k_forward 200 $O/timeline_e2e_rec.txt > /dev/null 2>&1
head -16 $O/kernel_stats_e2e_rec.txt | cut -c1-210; cat $O/timeline_e2e_rec.txt | cut -c1-160
find $O -name "*.db" -delete; rm -rf $O/prof

#!/bin/bash
R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd); cd $R
python - <<'PY'
import sys, os, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, 'tests')
from test_ingest import _criteo_text
from oracle import ingest as oi
text=_criteo_text(np.random.default_rng(21), 3000)
open('/tmp/t.criteo','wb').write(text)
off, lab, idx = oi.parse_criteo(text)
recs=[]
for a in range(0,3000,500):
    recs.append(oi.write_crb_record(off[a:a+501]-off[a], lab[a:a+500], idx[int(off[a]):int(off[a+500])]))
open('/tmp/t.rec','wb').write(oi.write_recordio(recs))
with open('/tmp/t.libsvm','w') as f:
    for i in range(3000):
        f.write("%d %s\n" % (lab[i], " ".join("%d:1" % v for v in idx[int(off[i]):int(off[i+1])])))
PY
run() { ./build/difacto data_in=/tmp/t.$1 data_format=$1 task=train learner=sgd batch_size=500 max_num_epochs=2 V_dim=4 V_threshold=0 l1=.01 lr=.1 V_lr=.05 V_init=hash table_capacity=262144 stop_rel_objv=0 $2 2>&1 | grep "Training: loss" | sed 's/.*loss = //; s/, AUC.*//' | tr '\n' ' '; echo " [$1 $2]"; }
export DIFACTO_TRACE=1
for f in criteo rec; do ./build/difacto data_in=/tmp/t.$f data_format=$f task=train learner=sgd batch_size=500 max_num_epochs=1 V_dim=4 V_threshold=0 l1=.01 lr=.1 V_lr=.05 V_init=hash table_capacity=262144 stop_rel_objv=0 2>&1 | grep "batch rows\|Training" | sed 's/.*INFO//' ; done

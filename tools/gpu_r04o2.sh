#!/bin/bash
# round 4: what the first minibatches of a job wait for (start-up of the worker loop): every dispatch of the first 60 ms
# of build/difacto on the 400 000-row .rec file, and the loop's profile lines
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04o2; mkdir -p $O; cd $R
E2E_KEEP=1 E2E_FORMATS=rec timeout 300 python tools/e2e_cli.py 400000 2 > /dev/null 2> $O/gen.err
F=$(ls -d /tmp/e2e_*/ | head -1)train.rec
ARGS="data_in=$F data_format=rec task=train learner=sgd batch_size=10000 max_num_epochs=1 V_dim=64 V_threshold=0 l1=0 lr=.01 V_lr=.01 V_init=hash table_capacity=8388608 stop_rel_objv=0 num_jobs_per_epoch=1"
DIFACTO_PROFILE=1 DFH_PROFILE_PREP=1 $R/build/difacto $ARGS 2>&1 | grep -E "host loop|reader: |prepare_rows x|load_host|batch reader" | sed 's/.*\] //' | cut -c1-300 | sort | uniq -c | sort -rn | head -12
cd /tmp
timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $O/prof -o kt -- $R/build/difacto $ARGS > $O/prof.log 2>&1
DB=$(ls $O/prof/*.db $O/prof/*/*.db 2>/dev/null | head -1)
python - "$DB" > $O/startup_timeline.txt <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = [(r[1], r[2], "K s%s %s" % (r[3], r[0][:70])) for r in db.execute("select name, start, end, stream_id from kernels")]
try:
    tabs = [t[0] for t in db.execute("select name from sqlite_master where type in ('table','view') and name like '%memory_cop%'")]
    for t in tabs[:1]:
        cols = [c[1] for c in db.execute("pragma table_info(%s)" % t)]
        if "start" in cols and "end" in cols:
            sz = "size" if "size" in cols else cols[0]
            for r in db.execute("select start, end, %s from %s" % (sz, t)):
                rows.append((r[0], r[1], "COPY %s B" % r[2]))
except Exception as e:
    print("# copies unavailable:", e)
rows.sort()
t0 = rows[0][0]
print("# every dispatch / copy of the first 60 ms after the first one; us")
for s, e, n in rows:
    if (s - t0) / 1e3 > 60000: break
    print("%10.1f %10.1f %8.1f  %s" % ((s - t0) / 1e3, (e - t0) / 1e3, (e - s) / 1e3, n))
PY
head -150 $O/startup_timeline.txt | cut -c1-130
rm -rf $O/prof

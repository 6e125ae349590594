#!/bin/bash
# round 6: where the end-to-end step goes — build/difacto on the 19.2 M-row .rec file under rocprofv3 --kernel-trace (per-kernel averages in
# the e2e condition: two preparation streams, uploads, the e2e generator's heavy-headed ids), fused gather against the gather launch,
# one against two preparation streams
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out/r06g && mkdir -p $O
export TMPDIR=/tmp
python -c "from difacto_amd.build import build_hip, build_host; build_hip(); build_host()" 2>&1 | tail -2
timeout 1500 python -m pytest tests/test_single_queue.py tests/test_nonfinite.py tests/test_shard_native.py tests/test_host_cpp.py -x -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 | tee $O/pytest_misc.txt
D=/tmp/e2e_prof; mkdir -p $D
python - <<'PY'
import numpy as np, os, sys
sys.path.insert(0, os.getcwd())
from oracle import ingest as oi
rows=400000; rng=np.random.default_rng(1)
ints = rng.zipf(1.3, size=(rows, 13)) % 10000
cats = (rng.zipf(1.1, size=(rows, 26)) % 1000000).astype(np.uint64) * np.uint64(2654435761) % np.uint64(1 << 32)
lab = (rng.random(rows) < 0.25).astype(np.int32)
lines = ["%d\t%s\t%s" % (lab[i], "\t".join(map(str, ints[i])), "\t".join("%08x" % c for c in cats[i])) for i in range(rows)]
text = ("\n".join(lines) + "\n").encode()
off, labf, idx = oi.parse_criteo(text)
recs = []
for a in range(0, rows, 10000):
    b = min(rows, a + 10000)
    recs.append(oi.write_crb_record(off[a:b + 1] - off[a], labf[a:b], idx[int(off[a]):int(off[b])]))
blob = oi.write_recordio(recs)
with open("/tmp/e2e_prof/train_x48.rec", "wb") as f:
    for _ in range(48): f.write(blob)
# unique keys per minibatch and the heaviest keys of the first one
k, c = np.unique(idx[:int(off[10000])], return_counts=True)
print("e2e generator, first minibatch: unique keys", len(k), "of", int(off[10000]), "pairs; keys with > 64 occurrences", int((c > 64).sum()), "; > 1000:", int((c > 1000).sum()), "; max", int(c.max()))
PY
ARGS="data_in=$D/train_x48.rec data_format=rec task=train learner=sgd batch_size=10000 max_num_epochs=1 V_dim=64 V_threshold=0 l1=0 lr=.01 V_lr=.01 V_init=hash table_capacity=8388608 stop_rel_objv=0 num_jobs_per_epoch=1"
for v in fused2 alone2 fused1 alone1; do
  case $v in fused2) E="";; alone2) E="DFH_GATHER_ALONE=1";; fused1) E="DIFACTO_PREP_STREAMS=1";; alone1) E="DFH_GATHER_ALONE=1 DIFACTO_PREP_STREAMS=1";; esac
  for rep in 1 2; do env $E DIFACTO_PROFILE=1 timeout 300 ./build/difacto $ARGS 2>&1 | grep "host loop over" | sed "s/^.*host loop/$v: host loop/" ; done
done | tee $O/e2e_variants.txt
cd /tmp
for v in fused2 alone2; do
  case $v in fused2) E="";; alone2) E="DFH_GATHER_ALONE=1";; esac
  env $E timeout 600 rocprofv3 --kernel-trace --stats -d $O/prof_$v -o kt -- $R/build/difacto $ARGS > $O/prof_$v.log 2>&1
  python $R/tools/rocpd_stats.py $(ls $O/prof_$v/*.db $O/prof_$v/*/*.db 2>/dev/null | head -1) $O/kernel_stats_e2e_rec_$v.txt > /dev/null 2>&1
  head -14 $O/kernel_stats_e2e_rec_$v.txt | cut -c1-200
done
find $O -name "*.db" -delete; rm -rf $O/prof_fused2 $O/prof_alone2

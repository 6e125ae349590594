#!/bin/bash
# round 5: the projection set — one GPU as rank r of W (loop-back transport): C4 at W = 8 (all ranks, then the heaviest in sync
# mode), W = 2, W = 4; kernel stats + HBM counters of the W = 8 step; C5 (1e9 ids, V_dim 128, blended ranges) at W = 8
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05f; mkdir -p $O; cd $R
run() { n=$1; shift; ( time timeout 900 python bench.py "$@" ) > $O/$n.json 2> $O/$n.err; tail -2 $O/$n.err | head -1
python - <<PY
import json
try:
    d=json.loads(open("$O/$n.json").read().strip().splitlines()[-1])
    print("$n", "proj M ex/s", {m:round(v/1e6,1) for m,v in d["projected_examples_per_sec"].items()}, "ranks", d["emulated_ranks"], d["config"].get("range_shares_per_owner"))
    for r in d["ranks"]:
        print("  rank", r["rank"], {m:round(v["ms_per_step"],4) for m,v in r["models"].items()}, "in/out", round(r["keys_in_per_batch"]), round(r["remote_keys_out_per_batch"]), "owned", r["owned_keys"], "stages(off)", r["stage_ms_per_step"].get("off"))
except Exception as e: print("$n ERR", e)
PY
}
run emul_c4_w8_all --emulate-world 8 --emulate-rank all --steps 50 --warmup 10 --min-time 1.0 --no-timing
run emul_c4_w8 --emulate-world 8 --steps 50 --warmup 10 --min-time 2.0
run emul_c4_w8_sync --emulate-world 8 --exchange sync --steps 50 --warmup 10 --min-time 2.0
run emul_c4_w4 --emulate-world 4 --steps 50 --warmup 10 --min-time 2.0
run emul_c4_w2 --emulate-world 2 --steps 50 --warmup 10 --min-time 2.0
run emul_c4_w8_ids --emulate-world 8 --key-ranges ids --steps 50 --warmup 10 --min-time 1.0 --no-timing
cd /tmp
E="--emulate-world 8 --steps 50 --warmup 10 --min-time 0.3 --no-timing"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_sync -o kt -- python $R/bench.py $E --exchange sync > $O/prof_sync.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_sync/*.db $O/prof_sync/*/*.db 2>/dev/null | head -1) $O/kernel_stats_emulated_w8_sync.txt > /dev/null 2>&1
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_ovl -o kt -- python $R/bench.py $E --exchange overlap > $O/prof_ovl.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_ovl/*.db $O/prof_ovl/*/*.db 2>/dev/null | head -1) $O/kernel_stats_emulated_w8_overlap.txt > /dev/null 2>&1
python $R/tools/rocpd_timeline.py $(ls $O/prof_ovl/*.db $O/prof_ovl/*/*.db 2>/dev/null | head -1) k_forward 5 $O/timeline_emulated_w8_overlap.txt > /dev/null 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $c --kernel-trace -d $O/pmc_$c -o pmc -- python $R/bench.py --emulate-world 8 --steps 10 --warmup 5 --min-time 0.001 --max-reps 1 --no-timing --exchange sync > $O/pmc_$c.log 2>&1
done
f() { ls $O/$1/*.db $O/$1/*/*.db 2>/dev/null | head -1; }
python $R/tools/pmc_summary.py $(f pmc_FETCH_SIZE) $(f pmc_WRITE_SIZE) $O/pmc_hbm_traffic_emulated_w8.json $O/pmc_hbm_traffic_emulated_w8.txt > /dev/null 2>&1
head -24 $O/kernel_stats_emulated_w8_sync.txt | cut -c1-160; head -20 $O/pmc_hbm_traffic_emulated_w8.txt
head -40 $O/timeline_emulated_w8_overlap.txt
find $O -name "*.db" -delete; rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE $O/prof_sync $O/prof_ovl
# C5: 1e9 ids, V_dim 128, l1 = 1, blended ranges, the heaviest owner; under the kernel trace (one expensive set-up)
timeout 1500 rocprofv3 --kernel-trace --stats -d $O/prof_c5 -o kt -- python $R/bench.py --preset c5-slice --ids 1000000000 --emulate-world 8 --distinct 16 --steps 50 --warmup 10 --min-time 1.5 > $O/emul_c5_w8.json 2> $O/emul_c5_w8.err
python $R/tools/rocpd_stats.py $(ls $O/prof_c5/*.db $O/prof_c5/*/*.db 2>/dev/null | head -1) $O/kernel_stats_emulated_c5_w8.txt > /dev/null 2>&1
head -22 $O/kernel_stats_emulated_c5_w8.txt | cut -c1-160; tail -3 $O/emul_c5_w8.err
cd $R; python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/emul_c5_w8.json").read().strip().splitlines() if l.startswith("{")][-1])
    print("c5", "proj M ex/s", {m:round(v/1e6,1) for m,v in d["projected_examples_per_sec"].items()}, "ranks", d["emulated_ranks"], d["config"].get("range_shares_per_owner"))
    for r in d["ranks"]:
        print("  rank", r["rank"], {m:round(v["ms_per_step"],4) for m,v in r["models"].items()}, "in/out", round(r["keys_in_per_batch"]), round(r["remote_keys_out_per_batch"]), "owned", r["owned_keys"], "prefill s", round(r["prefill_seconds"],1), "stages(off)", r["stage_ms_per_step"].get("off"))
except Exception as e: print("c5 ERR", e)
PY
find $O -name "*.db" -delete; rm -rf $O/prof_c5; du -sh $O

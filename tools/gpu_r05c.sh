#!/bin/bash
# round 5: linked-entry owner-side kernels + penalty folded into the gradient-row launch: parity, then the emulated rank again
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r05d; mkdir -p $O; cd $R
timeout 1200 python -m pytest tests/test_loopback_emulation.py tests/test_gpu_parity.py tests/test_shard_native.py tests/test_sharded_gpu_ranks.py tests/test_host_cpp.py -x -q -m gpu -k "loopback or owner_side or shard or sharded" > $O/pytest.txt 2>&1; tail -8 $O/pytest.txt
cd /tmp
E="--emulate-world 8 --emulate-rank 0 --steps 50 --warmup 10 --min-time 0.3 --no-timing"
timeout 400 rocprofv3 --kernel-trace --stats -d $O/prof_sync -o kt -- python $R/bench.py $E --exchange sync > $O/prof_sync.log 2>&1
python $R/tools/rocpd_stats.py $(ls $O/prof_sync/*.db $O/prof_sync/*/*.db 2>/dev/null | head -1) $O/kernel_stats_emulated_w8_sync.txt > /dev/null 2>&1
head -24 $O/kernel_stats_emulated_w8_sync.txt | cut -c1-180
cd $R
timeout 600 python bench.py --emulate-world 8 --emulate-rank 0 --steps 50 --warmup 10 --min-time 1.5 > $O/emul_w8_r0.json 2> $O/emul_w8_r0.err; tail -3 $O/emul_w8_r0.err
timeout 600 python bench.py --emulate-world 8 --emulate-rank 0 --exchange sync --steps 50 --warmup 10 --min-time 1.5 > $O/emul_w8_r0_sync.json 2> $O/emul_w8_r0_sync.err
python - <<'PY'
import json
for f in ("emul_w8_r0","emul_w8_r0_sync"):
    try:
        d=json.loads(open("gpurun_out/r05d/%s.json"%f).read().strip().splitlines()[-1])
        r=d["ranks"][0]
        print(f, {m:round(v["ms_per_step"],4) for m,v in r["models"].items()}, "keys in/out", r["keys_in_per_batch"], r["remote_keys_out_per_batch"])
        print("   stages off", r["stage_ms_per_step"].get("off"))
    except Exception as e: print(f, "ERR", e)
PY
find $O -name "*.db" -delete; rm -rf $O/prof_sync

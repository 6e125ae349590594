cd $GRAFT_REPO_ROOT; O=gpurun_out/r05y; mkdir -p $O
run() { n=$1; shift; timeout 900 python bench.py "$@" > $O/$n.json 2> $O/$n.err
python - <<PY
import json
try:
    d=json.loads([l for l in open("$O/$n.json").read().strip().splitlines() if l.startswith("{")][-1])
    print("$n", "proj M ex/s", {m:round(v/1e6,1) for m,v in d["projected_examples_per_sec"].items()}, "ranks", d["emulated_ranks"])
    for r in d["ranks"]:
        print("  rank", r["rank"], {m:round(v["ms_per_step"],4) for m,v in r["models"].items()})
except Exception as e: print("$n ERR", e)
PY
}
run emul_c4_w8_all --emulate-world 8 --emulate-rank all --steps 50 --warmup 10 --min-time 1.0 --no-timing
run emul_c4_w4 --emulate-world 4 --steps 50 --warmup 10 --min-time 2.0
run emul_c4_w2 --emulate-world 2 --steps 50 --warmup 10 --min-time 2.0
run emul_c5_w8 --preset c5-slice --ids 1000000000 --emulate-world 8 --distinct 16 --steps 50 --warmup 10 --min-time 1.5

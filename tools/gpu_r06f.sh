#!/bin/bash
# round 6: the row gather inside the Localizer's count pass (k_loc_count_gather) — parity, then build/difacto end to end with the
# gather as its own launch (DFH_GATHER_ALONE=1: rounds 3-5) against the fused pass, same files, same box; the wire probe in the gloo dry run
cd "$(dirname "$0")/.." && R=$PWD && O=$R/gpurun_out/r06f && mkdir -p $O
export TMPDIR=/tmp
python -c "from difacto_amd.build import build_hip, build_host; build_hip(); build_host()" 2>&1 | tail -2
timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "device_feed or device_row_gather or pipelined_prep" 2>&1 | tail -5 | tee $O/pytest_feed.txt
timeout 1200 python -m pytest tests/test_single_queue.py tests/test_nonfinite.py tests/test_shard_native.py tests/test_host_cpp.py -x -q -m gpu 2>&1 | tail -5 | tee $O/pytest_misc.txt
export DIFACTO_PROFILE=1
E2E_FORMATS=rec,criteo E2E_EXES=difacto@alone,difacto@fused E2E_VARIANTS="alone:DFH_GATHER_ALONE=1,fused:" timeout 1500 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python - $O/e2e.jsonl <<'PY'
import json,sys
for l in open(sys.argv[1]):
    d=json.loads(l)
    print(d["format"], d["exe"], "whole loop M rows/s %.1f steady %.1f loop_s small/big %.4f %.4f process wall small/big %.3f %.3f -> process M rows/s %.1f" % (
        d.get("loop_rows_per_s_big",0)/1e6, d.get("steady_rows_per_s_by_loop_clock",0)/1e6, d.get("loop_s",0), d.get("loop_s_big",0), d["wall_s"], d["wall_s_big"], d["rows_big"]/d["wall_s_big"]/1e6))
PY
grep -h "start-up\|host loop" $O/e2e.err | head -12
unset DIFACTO_PROFILE
# the N > 1 bench code on one GPU: 8 ranks over gloo, host-staged exchange (dry run) — the line must carry the wire probe
DFH_BENCH_BACKEND=gloo DFH_WIRE_PROBE_BYTES=100000,1000000 timeout 900 python bench.py --gpus 8 --steps 5 --warmup 2 --min-time 0 --ids 2000000 --rows 2000 --distinct 8 --cpu-batches 0 > $O/bench_dryrun_gloo_w8.json 2> $O/bench_dryrun_gloo_w8.err
tail -c 1500 $O/bench_dryrun_gloo_w8.json; echo
python -c "
import json
d=json.loads(open('$O/bench_dryrun_gloo_w8.json').read().strip().splitlines()[-1])
print('wire_probe', json.dumps(d.get('wire_probe'))[:600]); print('roofline_exchange', json.dumps(d.get('roofline_exchange'))[:500]); print('value', d.get('value'), 'lr_divided_by_world', d.get('lr_divided_by_world'))"

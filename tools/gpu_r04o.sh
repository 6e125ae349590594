#!/bin/bash
# device feed with the shuffle buffers uploaded as slices of the parsed chunks (no host assembly), against the host feed
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04o; mkdir -p $O; cd $R
sed -i 's/if "host loop over" in l or "reader: " in l or "batch reader" in l:/if "host loop over" in l or "reader: " in l or "batch reader" in l or "dfh_batch_prepare_rows" in l or "dfh_rowbuf_load_host" in l:/' tools/e2e_cli.py
( time timeout 600 python -m pytest tests -m gpu -q -k "row_gather or cli" ) > $O/pytest_gpu.log 2>&1
grep -E "passed|failed" $O/pytest_gpu.log | tail -2; grep -E "^FAILED|^E  " $O/pytest_gpu.log | head -20
E2E_FORMATS=criteo,rec E2E_VARIANTS="hostfeed:DIFACTO_HOST_FEED=1" E2E_EXES=difacto,difacto@hostfeed DIFACTO_PROFILE=1 DFH_PROFILE_PREP=1 timeout 900 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python -c "
import json
for l in open('$O/e2e.jsonl'):
    d=json.loads(l); print(d['format'], d['exe'], 'steady %.2f M rows/s' % (d['steady_rows_per_s']/1e6), 'big %.2f s' % d['wall_s_big'], 'small %.2f s' % d['wall_s'])"
grep -E "host loop over 1920|10000 rows, shuffle|100000 rows, shuffle|prepare_rows x 9|load_host x [0-9][0-9]" $O/e2e.err | cut -c1-260 | tail -40

#!/bin/bash
# device feed: parser threads 16 / 32 / 48 (slices uploaded by two threads, one buffer ahead)
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r04m; mkdir -p $O; cd $R
sed -i 's/if "host loop over" in l or "reader: " in l or "batch reader" in l:/if "host loop over" in l or "reader: " in l or "batch reader" in l or "dfh_batch_prepare_rows" in l or "dfh_rowbuf_load_host" in l:/' tools/e2e_cli.py
E2E_FORMATS=criteo,rec E2E_VARIANTS="p16:DIFACTO_PARSER_THREADS=16,p32:DIFACTO_PARSER_THREADS=32,p48:DIFACTO_PARSER_THREADS=48,p32u4:DIFACTO_PARSER_THREADS=32+DIFACTO_UPLOAD_THREADS=4" E2E_EXES=difacto@p16,difacto@p32,difacto@p48,difacto@p32u4 DIFACTO_PROFILE=1 timeout 1200 python tools/e2e_cli.py 400000 48 > $O/e2e.jsonl 2> $O/e2e.err
python -c "
import json
for l in open('$O/e2e.jsonl'):
    d=json.loads(l); print(d['format'], d['exe'], 'steady %.2f M rows/s' % (d['steady_rows_per_s']/1e6), 'big %.2f s' % d['wall_s_big'], 'small %.2f s' % d['wall_s'])"
grep -E "host loop over 1920|reader: (311|1920) chunks" $O/e2e.err | cut -c1-230 | tail -40

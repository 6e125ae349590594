#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r02l; mkdir -p $O; cd $R
( time timeout 900 python -m pytest tests -m gpu -x -q ) > $O/pytest_gpu.log 2>&1
tail -4 $O/pytest_gpu.log | cut -c1-300
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
timeout 300 ./build/difacto_host_tests tests/golden/rcv1_100.libsvm > $O/host_tests.log 2>&1; tail -3 $O/host_tests.log
timeout 300 python bench.py --cpu-batches 0 > $O/bench_c3.json 2> $O/bench_c3.err; python -c "
import json
d=json.loads(open('$O/bench_c3.json').read().strip().splitlines()[-1]); print(round(d['value']/1e6,2), d['ms_per_step'], d['kernel_ms_per_step'], d['roofline']['frac'])"

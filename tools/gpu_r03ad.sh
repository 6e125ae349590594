#!/bin/bash
export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; [ -z "$R" ] && R=$(pwd)
O=$R/gpurun_out/r03ad; mkdir -p $O; cd $R
python -c "from difacto_amd import build; build.build_host()" > $O/build.log 2>&1
sed -e 's/V_init = refrand/V_init = hash/' -e 's/max_num_epochs = 10/max_num_epochs = 2/' -e 's/batch_size = 100/batch_size = 25/' example/rcv1_fm.conf > /tmp/det.conf
for i in 1 2 3; do
  DIFACTO_TRACE=1 build/difacto argfile=/tmp/det.conf > $O/trace_$i.out 2> $O/trace_$i.err
  grep -E "batch rows|Training: loss" $O/trace_$i.err | sed -e 's/^.*\] //' > $O/trace_$i.txt
done
paste -d'|' $O/trace_1.txt $O/trace_2.txt | head -40 | cut -c1-260
cmp $O/trace_1.txt $O/trace_2.txt; cmp $O/trace_1.txt $O/trace_3.txt

#!/bin/bash
# registers / scratch / occupancy of the kernels whose mangled name matches $1 (regex), from -Rpass-analysis=kernel-resource-usage
# usage: tools/kernel_resources.sh 'k_forwardILi16ELi5|k_update_fusedILi16' [-DDFH_...]
R=$(cd "$(dirname "$0")/.." && pwd); pat=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -I$R/include -I$R/difacto_amd/csrc "$@" \
  -Rpass-analysis=kernel-resource-usage -o /tmp/kr_$$.so $R/difacto_amd/csrc/dfh_api.hip 2> /tmp/kr_$$.txt
python3 - "$pat" /tmp/kr_$$.txt <<'PY'
import re,sys
pat=re.compile(sys.argv[1]); t=open(sys.argv[2]).read()
for b in re.split(r'(?=remark: [^\n]*Function Name:)',t):
    m=re.search(r'Function Name: (\S+)',b)
    if not m or not pat.search(m.group(1)): continue
    g=lambda k:(re.search(k+r': (\S+)',b) or [None,'?'])[1]
    print(m.group(1)[:72],'VGPR',g('VGPRs'),'SGPR',g('SGPRs'),'scratch',g(r'ScratchSize \[bytes/lane\]'),'occ',g(r'Occupancy \[waves/SIMD\]'),'LDS',g(r'LDS Size \[bytes/block\]'))
PY
rm -f /tmp/kr_$$.so /tmp/kr_$$.txt

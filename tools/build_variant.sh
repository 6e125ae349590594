#!/bin/bash
# build a variant of the product library for same-box A/B runs: tools/var_<name>.so (git-ignored; travels with gpurun)
# usage: tools/build_variant.sh <name> [-DDFH_...=... ...]
R=$(cd "$(dirname "$0")/.." && pwd); n=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -Wno-unused-value -I$R/include -I$R/difacto_amd/csrc "$@" \
  -o $R/tools/var_$n.so $R/difacto_amd/csrc/dfh_api.hip

#!/bin/bash
# k_nb_ems2 built with different -D switches (run ON the GPU box, repo root): bench rate of each variant.
# usage: bash tools/nb_pipe_variants.sh "-DNB_PIPE_WALK_COLS=2" "-DNB_PIPE_WALK_COLS=4" ...
set -e -o pipefail
mkdir -p gpurun_out/stamp
FL="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I include"
i=0
for V in "$@"; do
  i=$((i+1))
  hipcc $FL $V -c cuda_ldpc_amd/csrc/nbldpc_api.hip -o gpurun_out/stamp/nb_v$i.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/stamp/lib_v$i.so gpurun_out/stamp/nb_v$i.o build/obj/bldpc_api.o build/obj/bldpc_channel.o
  echo "== $V"
  CUDA_LDPC_AMD_SO=$PWD/gpurun_out/stamp/lib_v$i.so python bench.py --workload NB_BDS_GF64 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;j=json.loads(sys.stdin.read());print('bench', j['value'], j['roofline']['kernel_ms'])"
done
rm -rf gpurun_out/stamp

#!/bin/bash
# k_nb_ems2 with wave priorities in stage 2 (run ON the GPU box): NB_PIPE_PRIO 0 / 1 / 2 builds, stamps and bench rate of each.
set -e -o pipefail
mkdir -p gpurun_out/stamp
FL="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I include"
for P in 0 1 2; do
  hipcc $FL -DNB_STAMP=1 -DNB_PIPE_PRIO=$P -c cuda_ldpc_amd/csrc/nbldpc_api.hip -o gpurun_out/stamp/nb_p$P.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/stamp/libstamp_p$P.so gpurun_out/stamp/nb_p$P.o build/obj/bldpc_api.o build/obj/bldpc_channel.o
  hipcc $FL -DNB_PIPE_PRIO=$P -c cuda_ldpc_amd/csrc/nbldpc_api.hip -o gpurun_out/stamp/nb_q$P.o
  hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/stamp/lib_p$P.so gpurun_out/stamp/nb_q$P.o build/obj/bldpc_api.o build/obj/bldpc_channel.o
  echo "== NB_PIPE_PRIO=$P"
  CUDA_LDPC_AMD_SO=$PWD/gpurun_out/stamp/libstamp_p$P.so python tools/nb_pipe_stamp.py 2>&1 | grep "B=4096"
  CUDA_LDPC_AMD_SO=$PWD/gpurun_out/stamp/lib_p$P.so python bench.py --workload NB_BDS_GF64 --no-cpu-baseline 2>/dev/null | python -c "import json,sys;j=json.loads(sys.stdin.read());print('bench', j['value'], j['roofline']['kernel_ms'])"
done

#!/bin/bash
# times the J15_L30_Z1280 decode with each ablation build of build/ablate/lib<n>.so (the library compiled with -DQCR2_ABLATE=<n>:
# wrong results, timing only; see bldpc_qcr2_kernel.hpp), then the product, then k_qcr (BLDPC_NO_HALO=1)
for n in 3 4 8 16; do
  echo "== QCR2_ABLATE=$n"; CUDA_LDPC_AMD_SO=$PWD/build/ablate/lib$n.so timeout -k 10 120 python bench.py --workload J15_L30_Z1280 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*'
done
echo "== product"; timeout -k 10 120 python bench.py --workload J15_L30_Z1280 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*\|regstate[a-z0-9-]*'
echo "== old k_qcr"; BLDPC_NO_HALO=1 timeout -k 10 120 python bench.py --workload J15_L30_Z1280 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | grep -o '"kernel_ms": [0-9.]*\|regstate[a-z0-9-]*'

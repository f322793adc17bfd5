#!/bin/bash
# Experiment driver (GPU box): bench.py's default workload on the experiment builds under build/exp/<name>/lib.so (CUDA_LDPC_AMD_SO),
# three runs each interleaved with the shipped library, then the phase probe.  usage: bash tools/variant_bench.sh name...
for rep in 1 2; do
  for v in base "$@"; do
    if [ $v = base ]; then so=; else so=$PWD/build/exp/$v/lib.so; fi
    CUDA_LDPC_AMD_SO=$so python bench.py --no-cpu-baseline ${WL:+--workload $WL} 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', '%.3f M cw/s' % (j['value']/1e6), 'kernel %.3f ms' % j['roofline']['kernel_ms'])"
  done
done

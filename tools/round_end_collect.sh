#!/bin/bash
# Round-end collection (run ON the GPU box, repo root): full GPU suite, smoke, default bench, the secondary NB bench lines, workspace-kernel
# timing and the shim sweeps; summaries land in gpurun_out/profiles_out/ (copy them into profiles/).
set -o pipefail
mkdir -p gpurun_out/profiles_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/t_full.log 2>&1; tail -3 gpurun_out/t_full.log
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
python bench.py > gpurun_out/profiles_out/default_bench.json 2> gpurun_out/default_bench.err; python -c "import json;j=json.load(open('gpurun_out/profiles_out/default_bench.json'));print('default', j['value'], j['roofline']['frac'], j['cpu_baseline']['value'])"
for w in NB_N96_GF256 NB_BDS_GF64_TMM NB_BDS_GF64_LTMM; do python bench.py --workload $w > gpurun_out/profiles_out/r03_${w}_bench.json 2>/dev/null; python -c "import json;j=json.load(open('gpurun_out/profiles_out/r03_${w}_bench.json'));print('$w', j['value'])"; done
python tools/nb_hbm_time.py > gpurun_out/profiles_out/r03_nb_hbm_time.txt 2>&1; tail -2 gpurun_out/profiles_out/r03_nb_hbm_time.txt
bash tools/shim_sweep_time.sh > gpurun_out/profiles_out/r03_shim_sweep.txt 2>&1; grep "^real" gpurun_out/profiles_out/r03_shim_sweep.txt
rm -rf gpurun_out/shim_bin

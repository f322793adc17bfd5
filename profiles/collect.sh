#!/bin/bash
# Collect the rocprofv3 evidence behind one bench line (run ON the GPU box, from the repo root):
#     bash profiles/collect.sh <tag> <name> <kernel-match> [bench.py flags...]
# Writes gpurun_out/<tag>_<name>_bench.json (plain run) and four profiled runs of the SAME command --
# kernel trace + stats, then FETCH_SIZE, WRITE_SIZE and the SQ counters in SEPARATE --pmc passes (counters are never
# combined with tracing domains) -- and condenses them into profiles/<tag>_<name>_{kernel_stats.csv,pmc.json}.
set -e -o pipefail
tag=$1; name=$2; match=$3; shift 3
out=gpurun_out/${tag}_${name}
export TMPDIR=/tmp
python bench.py "$@" > ${out}_bench.json 2> ${out}_bench.err
rocprofv3 --kernel-trace --stats -d ${out}_trace -o run -- python bench.py "$@" --no-cpu-baseline --steps 20 --warmup 2 > ${out}_trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d ${out}_fetch -o run -- python bench.py "$@" --no-cpu-baseline --steps 2 --warmup 1 > ${out}_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d ${out}_write -o run -- python bench.py "$@" --no-cpu-baseline --steps 2 --warmup 1 > ${out}_write.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_WAVE_CYCLES -d ${out}_sq -o run -- python bench.py "$@" --no-cpu-baseline --steps 2 --warmup 1 > ${out}_sq.log 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY -d ${out}_sq2 -o run -- python bench.py "$@" --no-cpu-baseline --steps 2 --warmup 1 > ${out}_sq2.log 2>&1
# the wave-cycle split of MI355X_MICROARCH.md (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES)
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES -d ${out}_sq3 -o run -- python bench.py "$@" --no-cpu-baseline --steps 2 --warmup 1 > ${out}_sq3.log 2>&1
python profiles/summarize.py $tag $name ${out}_trace ${out}_fetch ${out}_write ${out}_sq,${out}_sq2,${out}_sq3 "$match"
cp ${out}_bench.json profiles/${tag}_${name}_bench.json
# gpurun merges back gpurun_out/ only: the summaries travel home in gpurun_out/profiles_out/ (copy them into profiles/ there)
mkdir -p gpurun_out/profiles_out
cp profiles/${tag}_${name}_bench.json profiles/${tag}_${name}_kernel_stats.csv profiles/${tag}_${name}_pmc.json gpurun_out/profiles_out/
rm -rf ${out}_trace ${out}_fetch ${out}_write ${out}_sq ${out}_sq2 ${out}_sq3

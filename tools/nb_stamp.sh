#!/bin/bash
# Phase stamps of k_nb_ems (run ON the GPU box, repo root): builds the library with -DNB_STAMP into gpurun_out/ and runs tools/nb_stamp.py on it.
set -e -o pipefail
mkdir -p gpurun_out/stamp
FL="-O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function -I include"
hipcc $FL -DNB_STAMP=1 -c cuda_ldpc_amd/csrc/nbldpc_api.hip -o gpurun_out/stamp/nbldpc_api.o
hipcc --offload-arch=gfx950 -shared -fPIC -o gpurun_out/stamp/libstamp.so gpurun_out/stamp/nbldpc_api.o build/obj/bldpc_api.o build/obj/bldpc_channel.o
CUDA_LDPC_AMD_SO=$PWD/gpurun_out/stamp/libstamp.so NBLDPC_NO_PIPE=1 python tools/nb_stamp.py
CUDA_LDPC_AMD_SO=$PWD/gpurun_out/stamp/libstamp.so python tools/nb_pipe_stamp.py

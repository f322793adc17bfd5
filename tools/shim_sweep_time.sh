#!/bin/bash
# Timing of the reference-signature C++ harnesses in their fast modes next to the Python sweeps (run ON the GPU box, repo root):
#     bash tools/shim_sweep_time.sh > profiles/r03_shim_sweep.txt
# binary: tests/cpp/ref_main_style_sweep.cpp through shim/ldpc_ref_shim.hip with bldpc_shim_configure_fast(1, 1, PER_FRAME, 1200)
#         vs  sweep.py binary --device-channel --per-frame  (the sweep of profiles/r02c_sweep_binary_J4_L24_Z96_per_frame_exit.txt)
# nb:     tests/cpp/nb_ref_main_style_sweep.cpp through shim/nbldpc_ref_shim.hip, host channel (bit-exact stream) and device channel
#         vs  sweep.py nb [--device-channel]
set -e -o pipefail
R=$(pwd)
B=gpurun_out/shim_bin; mkdir -p $B
FL="-O2 --offload-arch=gfx950 -std=c++17 -I include -I shim -L cuda_ldpc_amd -lcuda_ldpc_amd -Wl,-rpath,$R/cuda_ldpc_amd -pthread"
hipcc $FL tests/cpp/ref_main_style_sweep.cpp shim/ldpc_ref_shim.hip -o $B/bmain
hipcc $FL tests/cpp/nb_ref_main_style_sweep.cpp shim/nbldpc_ref_shim.hip -o $B/nbmain
M=data/bldpc/J4_L24_Z96_BlockH.txt
echo "== binary J4_L24_Z96, per-frame exit, device channel, batch 262144, <= 1200 batches per point, Es/N0 3.0 ... 4.8"
echo "-- C++ harness (Simulation_GPU with the reference's signature, shim fast path)"
( time $B/bmain $M 4 24 96 262144 50 0 3.0 4.81 0.2 50 10000 1 1 2 1200 ) 2>&1 | grep -v "^POINT\|^$"
echo "-- sweep.py binary --device-channel --per-frame (Python mirror)"
( time python sweep.py binary --matrix $M --J 4 --L 24 --Z 96 --start 3.0 --stop 4.81 --step 0.2 --batch 262144 --max-batches 1200 --per-frame --device-channel ) 2>&1 | grep -v amdgpu.ids
echo "== binary, the reference's own data flow through the same entry point (host channel, D copied back, host Statistic): batch 4096, 3.0 dB"
( time $B/bmain $M 4 24 96 4096 50 0 3.0 3.01 0.2 50 10000 ) 2>&1 | grep -v "^POINT\|^$"
cd data/nb
echo "== NB BDS GF(64) EMS(2,2), Eb/N0 2.0 ... 4.0, the reference's stop rule (50 error frames, 1000 frames)"
echo "-- C++ harness, Simulation_GPU, host channel (the reference's bit-exact noise stream), batch 4096"
( time $R/$B/nbmain BDS.576.288.GF.64.txt codeword_bds_gf64.txt 64 4 2 0 1 2.0 4.01 0.5 50 1000 4096 0 ) 2>&1 | grep -v "^$"
echo "-- C++ harness, Simulation_GPU, device channel, batch 16384"
( time $R/$B/nbmain BDS.576.288.GF.64.txt codeword_bds_gf64.txt 64 4 2 0 1 2.0 4.01 0.5 50 1000 16384 1 ) 2>&1 | grep -v "^$"
cd $R
echo "-- sweep.py nb --device-channel --batch 16384"
( time python sweep.py nb --start 2.0 --stop 4.01 --step 0.5 --batch 16384 --device-channel ) 2>&1 | grep -v amdgpu.ids
echo "-- sweep.py nb (host channel, batch 1024)"
( time python sweep.py nb --start 2.0 --stop 4.01 --step 0.5 ) 2>&1 | grep -v amdgpu.ids

bash profiles/collect.sh r03 binary "k_qc2<" > gpurun_out/collect_binary.log 2>&1; tail -2 gpurun_out/collect_binary.log
bash profiles/collect.sh r03 j32 "k_qc<" --workload J32_L64_Z64 > gpurun_out/collect_j32.log 2>&1; tail -2 gpurun_out/collect_j32.log
./build/qc_phase_probe 65536 > gpurun_out/profiles_out/r03_qc2_phase_probe.txt 2>&1; BLDPC_NO_LOCAL=1 ./build/qc_phase_probe 65536 >> gpurun_out/profiles_out/r03_qc2_phase_probe.txt 2>&1
./build/qc_phase_probe 32768 0 32 64 64 data/bldpc/J32_L64_Z64_BlockH.txt > gpurun_out/profiles_out/r03_qc_j32_phase_probe.txt 2>&1; BLDPC_NO_LOCAL=1 ./build/qc_phase_probe 32768 0 32 64 64 data/bldpc/J32_L64_Z64_BlockH.txt >> gpurun_out/profiles_out/r03_qc_j32_phase_probe.txt 2>&1
grep -h "^kernel\|median\|mean iter" gpurun_out/profiles_out/*probe.txt | cut -c1-160
python tools/per_frame_time.py > gpurun_out/profiles_out/r03_per_frame_exit_throughput.txt 2>&1; tail -12 gpurun_out/profiles_out/r03_per_frame_exit_throughput.txt | cut -c1-200

# round 3: MFMA-shape lab on the headline shapes, random operands then zeros (tools/h3_mfma16_lab.hip)
cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o /tmp/h3_mfma16_lab tools/h3_mfma16_lab.hip || exit 1
LAB_R3=1 timeout -k 10 500 /tmp/h3_mfma16_lab > gpurun_out/lab_mfma16_random.log 2>&1
LAB_R3=1 LAB_ZERO=1 timeout -k 10 300 /tmp/h3_mfma16_lab > gpurun_out/lab_mfma16_zero.log 2>&1
grep -h "us " gpurun_out/lab_mfma16_random.log gpurun_out/lab_mfma16_zero.log

# Diagnostic library for tools/stamp_step.py: csrc/rows.hip and csrc/pointwise.hip compiled with -DROWS_STAMP=1 (wall-clock
# stamps at the phase boundaries of the few-row kernels and of the beam select), linked with the shipped objects of the
# other sources.  Writes tools/_lab/stamp/libinsenticap_hip_stamp.so (git-ignored; travels to the GPU box).
#   bash tools/r4_stamp_build.sh && gpurun -- 'ISC_HIP_LIB=tools/_lab/stamp/libinsenticap_hip_stamp.so python tools/stamp_step.py'
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/insenticap_model_amd/lib
O=$R/tools/_lab/stamp
mkdir -p $O
for f in rows pointwise; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DROWS_STAMP=1 -I$R/include -c $R/insenticap_model_amd/csrc/$f.hip -o $O/$f.o
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $O/libinsenticap_hip_stamp.so $L/gemm_f32.o $L/attention.o $O/pointwise.o $L/backward.o $L/step.o $O/rows.o
ls -la $O/libinsenticap_hip_stamp.so

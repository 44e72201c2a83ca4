# Where the vocabulary kernel's time goes (round-3 verdict item 4): a diagnostic build of csrc/gemm_f32.hip with
# -DH3_STAMP=1 puts s_memtime stamps around gemm_h3_kernel's setup, main loop and epilogue (summed over the waves of every
# launch of the vocabulary form); tools/h3_stamp.py runs greedy roll-outs at B = 4096 through it and prints the shares.
#   usage: bash tools/h3_stamp.sh build   (anywhere with hipcc: writes tools/_lab/libisc_stamp.so)
#          bash tools/h3_stamp.sh run     (on the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/insenticap_model_amd/lib
mkdir -p $R/tools/_lab
if [ "$1" = build ]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DH3_STAMP=1 -c $R/insenticap_model_amd/csrc/gemm_f32.hip -o /tmp/gemm_stamp.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_lab/libisc_stamp.so /tmp/gemm_stamp.o $L/attention.o $L/pointwise.o $L/backward.o $L/step.o $L/rows.o
  exit 0
fi
ISC_HIP_LIB=$R/tools/_lab/libisc_stamp.so timeout -k 10 300 python3 $R/tools/h3_stamp.py

# Where the time of the large split-f16 kernels goes: a diagnostic build of csrc/gemm_f32.hip with -DH3_STAMP=1 puts
# s_memtime stamps around the setup, main loop and epilogue of gemm_h3_kernel<vocab> and of gemm_h3x_kernel's LSTM / linear
# forms (summed over the waves of every launch); tools/h3_stamp.py runs greedy roll-outs at B = 16384 / 4096 through it and
# prints the shares.
#   usage: bash tools/h3_stamp.sh build [name [extra hipcc flags]]   (anywhere with hipcc: writes tools/_lab/libisc_<name>.so,
#                                                                     name defaults to "stamp")
#          bash tools/h3_stamp.sh run [name ...]                     (on the GPU box)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/insenticap_model_amd/lib
mkdir -p $R/tools/_lab
if [ "$1" = build ]; then
  N=${2:-stamp}
  shift; shift || true
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DH3_STAMP=1 "$@" -I$R/include -c $R/insenticap_model_amd/csrc/gemm_f32.hip -o /tmp/gemm_$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_lab/libisc_$N.so /tmp/gemm_$N.o $L/attention.o $L/pointwise.o $L/backward.o $L/step.o $L/rows.o
  exit 0
fi
shift
for N in ${@:-stamp}; do
  echo "== $N"
  ISC_HIP_LIB=$R/tools/_lab/libisc_$N.so timeout -k 10 300 python3 $R/tools/h3_stamp.py
done

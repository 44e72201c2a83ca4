# Where the skinny split-f16 kernels' time goes: builds three timing-only variants of
# csrc/gemm_f32.hip (-DH3S_LAB_MODE=1 DMA only, =2 DMA + fragment reads, =3 DMA + MFMAs on constant fragments; results are
# WRONG in all three), swaps each in for the library, and times the few-row entry points with tools/skinny_bench.py.
#   usage: bash tools/h3s_lab_modes.sh build      (anywhere with hipcc: writes tools/_lab/libisc_lab{1,2,3}.so)
#          bash tools/h3s_lab_modes.sh run        (on the GPU box: prints skinny_bench lines per variant, restores the library)
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
L=$R/insenticap_model_amd/lib
mkdir -p $R/tools/_lab
if [ "$1" = build ]; then
  for m in 1 2 3; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -DH3S_LAB_MODE=$m -c $R/insenticap_model_amd/csrc/gemm_f32.hip -o /tmp/gemm_lab$m.o
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/tools/_lab/libisc_lab$m.so /tmp/gemm_lab$m.o $L/attention.o $L/pointwise.o $L/backward.o $L/step.o
  done
  exit 0
fi
cp $L/libinsenticap_hip.so /tmp/libinsenticap_hip.orig.so
trap 'cp /tmp/libinsenticap_hip.orig.so $L/libinsenticap_hip.so' EXIT
for m in 0 1 2 3; do
  echo "== H3S_LAB_MODE $m"
  if [ $m != 0 ]; then cp $R/tools/_lab/libisc_lab$m.so $L/libinsenticap_hip.so; else cp /tmp/libinsenticap_hip.orig.so $L/libinsenticap_hip.so; fi
  timeout -k 10 200 python3 $R/tools/skinny_bench.py --mode 4 --rows 512,1024
  timeout -k 10 200 python3 $R/tools/skinny_bench.py --mode 3 --rows 512,1024
done

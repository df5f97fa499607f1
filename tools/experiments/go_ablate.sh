#!/bin/bash
# Timing ablations of csrc/winograd_gemm_out.hip on one layer (results are wrong under ablation; only the time is read).
# usage: tools/go_ablate.sh [layer=c3_2] ; builds variant libraries under tools/scratch/go_abl/
cd "$(dirname "$0")/.."
L=${1:-c3_2}; D=tools/scratch/go_abl; mkdir -p $D
C=strotss-tensorflow_amd/csrc
for a in BASE NO_DMA NO_MFMA NO_FOLD NO_EPI "NO_FOLD -DGO_ABL_NO_EPI" "NO_DMA -DGO_ABL_NO_FOLD -DGO_ABL_NO_EPI"; do
  tag=$(echo "$a" | tr -d ' -' | sed 's/DGO_ABL_/_/g')
  if [ ! -e $D/lib_$tag.so ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -w -DGO_ABL_$a -I $C -c $C/winograd_gemm_out.hip -o $D/go_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_$tag.so $C/gemm.o $C/losses.o $C/conv.o $C/winograd.o $C/winograd_fused.o $D/go_$tag.o $C/image.o
  fi
  [ "$BUILD_ONLY" = 1 ] && continue
  echo "== $tag"; STROTSS_HIP_LIB=$PWD/$D/lib_$tag.so STROTSS_WINOGRAD_TILE=4 STROTSS_WINO_GEMM_OUT=2 python3 tools/conv_bench.py 1024 20 $L 2>&1 | grep -E "wfwd|wdgrad"
done

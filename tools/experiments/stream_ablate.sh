#!/bin/bash
# Timing ablations of csrc/mfma_x3_stream.h on one layer (results are wrong under ablation; only the time is read).
cd "$(dirname "$0")/.."
L=${1:-c3_2}; D=tools/scratch/stream_abl; mkdir -p $D
C=strotss-tensorflow_amd/csrc
for a in BASE NO_STORE NO_DMA NO_MFMA "NO_STORE -DX3_ABL_NO_DMA"; do
  tag=$(echo "$a" | tr -d ' -' | sed 's/DX3_ABL_/_/g')
  if [ ! -e $D/lib_$tag.so ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=on -w -DX3_ABL_$a -I $C -c $C/gemm.hip -o $D/gemm_$tag.o &&
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $D/lib_$tag.so $D/gemm_$tag.o $C/losses.o $C/conv.o $C/winograd.o $C/winograd_fused.o $C/winograd_gemm_out.o $C/image.o
  fi
  [ "$BUILD_ONLY" = 1 ] && continue
  echo "== $tag"; STROTSS_HIP_LIB=$PWD/$D/lib_$tag.so STROTSS_WINOGRAD_TILE=4 STROTSS_X3_STREAM=1 python3 tools/conv_bench.py 1024 20 $L 2>&1 | grep -E "wfwd|wdgrad"
done

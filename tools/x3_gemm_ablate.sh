#!/bin/bash
# builds and runs tools/x3_gemm_ablate.hip for a list of ablations: tools/x3_gemm_ablate.sh [M N K batch]
for abl in "" "-DX3_ABL_NO_DMA" "-DX3_ABL_NO_FRAG" "-DX3_ABL_NO_MFMA" "-DX3_ABL_NO_BARRIER" "-DX3_ABL_NO_DMA -DX3_ABL_NO_FRAG" \
           "-DX3_ABL_NO_DMA -DX3_ABL_NO_FRAG -DX3_ABL_NO_BARRIER" "-DX3_ABL_NO_MFMA -DX3_ABL_NO_FRAG" "-DTILE=64" "-DUSE_K16" "-DUSE_K16 -DX3_ABL_NO_DMA" "-DUSE_K16 -DX3_ABL_NO_MFMA" "-DUSE_K16 -DX3_ABL_NO_DMA -DX3_ABL_NO_FRAG -DX3_ABL_NO_BARRIER"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -w $abl -I strotss-tensorflow_amd/csrc tools/x3_gemm_ablate.hip -o /tmp/x3g || exit 1
  echo "[${abl:-full}] $(/tmp/x3g "$@")"
done

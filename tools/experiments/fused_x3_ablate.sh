#!/bin/bash
# builds and runs tools/fused_x3_ablate.hip for a list of ablations: tools/fused_x3_ablate.sh [hw cin cout]
for abl in "" "-DX3_NO_MFMA" "-DX3_NO_U" "-DX3_NO_RAW" "-DX3_NO_SPLIT" "-DX3_NO_VSTORE" "-DX3_NO_SPLIT -DX3_NO_VSTORE" "-DX3_NO_TRANSFORM" \
           "-DX3_NO_TRANSFORM -DX3_NO_MFMA" "-DX3_NO_TRANSFORM -DX3_NO_U -DX3_NO_RAW" "-DX3_NO_TRANSFORM -DX3_NO_MFMA -DX3_NO_U -DX3_NO_RAW"; do
  hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=on -w $abl -I strotss-tensorflow_amd/csrc tools/fused_x3_ablate.hip -o /tmp/fx3 || exit 1
  echo "[${abl:-full}] $(/tmp/fx3 "$@")"
done

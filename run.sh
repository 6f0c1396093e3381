#!/bin/bash
# Usage (same as the reference's run.sh):  ./run.sh main <dataset> <mode> <epochs> [--no-build]
#   dataset: mnist | bloodmnist | fashionmnist      mode: train | predict
# Builds libvtcnn_hip.so + cpp/build/main with hipcc for gfx950 and runs it on HIP device 0.
set -e
HERE="$(cd "$(dirname "$0")" && pwd)"
EXE="$1"; DATASET="$2"; MODE="$3"; EPOCHS="$4"
NO_BUILD=0
for a in "$@"; do [ "$a" = "--no-build" ] && NO_BUILD=1; done
if [ -z "$EXE" ] || [ -z "$DATASET" ] || [ -z "$MODE" ]; then
  echo "Uso: ./run.sh main <mnist|bloodmnist|fashionmnist> <train|predict> <epocas> [--no-build]"; exit 1
fi
if [ "$NO_BUILD" = "0" ]; then
  make -s -j4 -C "$HERE/vit_amd/csrc"
  make -s -C "$HERE/cpp"
fi
if [ "$MODE" = "train" ]; then
  time "$HERE/cpp/build/$EXE" "$DATASET" "$MODE" "${EPOCHS:-5}"
else
  time "$HERE/cpp/build/$EXE" "$DATASET" "$MODE"
fi

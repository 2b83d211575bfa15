#!/bin/bash
# Build the standalone kernel probes against the library's object files (run after `python -m sam_road_amd.build`).
set -e
cd "$(dirname "$0")/../.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
for p in gemm_probe; do
  $HIPCC --offload-arch=gfx950 -O3 -std=c++17 -c tools/probes/$p.hip -o sam_road_amd/build/$p.o
  $HIPCC --offload-arch=gfx950 sam_road_amd/build/$p.o sam_road_amd/build/gemm.o sam_road_amd/build/gemm_q192.o -o tools/probes/$p
done
for p in feed_probe pipe_probe mfma_probe dma_probe; do
  [ -f tools/probes/$p.hip ] && $HIPCC --offload-arch=gfx950 -O3 -std=c++17 tools/probes/$p.hip -o tools/probes/$p
done

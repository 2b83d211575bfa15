#!/bin/bash
# Build the standalone kernel probes.  The GEMM probe links its OWN objects of the product's GEMM sources compiled with
# -DSRH_TUNING -Itools/probes, which pulls in tools/probes/gemm_tuning.inc / attn_tuning.inc: the probe-only kernels and the kernel / ablation
# selection by number (SRH_GEMM_VARIANT, GemmParams::variant, SRH_ATTN_ABL), the q192
# schedule variants and SRH_Q192_GR / SRH_Q192_PRIO — none of which exist in libsamroad_hip.so.
set -e
cd "$(dirname "$0")/../.."
HIPCC=${HIPCC:-/opt/rocm/bin/hipcc}
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -DSRH_TUNING -Itools/probes/build -Itools/probes"
mkdir -p tools/probes/build
python3 tools/kgen/gemm_z192_gen.py --variants tools/probes/build   # z192_var{k}_act{0,1}.inc + z192_var_kernels.inc: the schedule variants under test
python3 tools/kgen/attn_g64_gen.py                                  # attn_g64_body.inc / _meta.inc: the asm global attention experiment (probe only)
python3 tools/kgen/attn_g64_gen.py --variants tools/probes/build    # attn_g64_var{1..8}.inc: its schedule ablations
mkdir -p tools/probes/build
for f in gemm gemm_z192; do
  $HIPCC $FLAGS -c sam_road_amd/csrc/$f.hip -o tools/probes/build/$f.o &
done
$HIPCC $FLAGS -Isam_road_amd/csrc -c tools/probes/gemm_q192.hip -o tools/probes/build/gemm_q192.o &      # z192's predecessor: probe builds only (A/B history)
$HIPCC $FLAGS -c tools/probes/gemm_probe.hip -o tools/probes/build/gemm_probe.o &
wait
$HIPCC --offload-arch=gfx950 tools/probes/build/gemm_probe.o tools/probes/build/gemm.o tools/probes/build/gemm_q192.o tools/probes/build/gemm_z192.o -o tools/probes/gemm_probe
# windowed-attention probe: old vs new kernel of attention.hip (SRH_TUNING: the ablation switches and the old kernel behind ablate += 16)
for f in attention attention_hdx; do
  $HIPCC $FLAGS -DSRH_G64_PROBE -fno-honor-nans -c sam_road_amd/csrc/$f.hip -o tools/probes/build/$f.o &
done
$HIPCC $FLAGS -c tools/probes/attention_g64.hip -o tools/probes/build/attention_g64.o &
$HIPCC $FLAGS -c tools/probes/attn_win_probe.hip -o tools/probes/build/attn_win_probe.o &
wait
$HIPCC --offload-arch=gfx950 tools/probes/build/attn_win_probe.o tools/probes/build/attention.o tools/probes/build/attention_g64.o tools/probes/build/attention_hdx.o -o tools/probes/attn_win_probe
# head-dim-80 attention (ViT-H) alone, with its ablations
$HIPCC $FLAGS -c tools/probes/hdx_probe.hip -o tools/probes/build/hdx_probe.o
$HIPCC --offload-arch=gfx950 tools/probes/build/hdx_probe.o tools/probes/build/attention_hdx.o -o tools/probes/hdx_probe
# the fused map_decoder alone: batch sweep, with and without its stores
$HIPCC $FLAGS -c sam_road_amd/csrc/decoder.hip -o tools/probes/build/decoder.o
$HIPCC $FLAGS -c tools/probes/decoder_probe.hip -o tools/probes/build/decoder_probe.o
$HIPCC --offload-arch=gfx950 tools/probes/build/decoder_probe.o tools/probes/build/decoder.o -o tools/probes/decoder_probe
for p in feed_probe pipe_probe mfma_probe dma_probe feedx_probe mfma_data_probe store_probe trans_probe ln_probe; do
  [ -f tools/probes/$p.hip ] && $HIPCC --offload-arch=gfx950 -O3 -std=c++17 tools/probes/$p.hip -o tools/probes/$p
done

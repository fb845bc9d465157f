#!/bin/bash
# Round-6 measurement pass on the GPU box (one gpurun call): the tests of everything this round touched, the A-B lines of config 4 and the full NeRFPlayer,
# and the config-4 / full-NeRFPlayer profiles.  usage: bash tools/r06_measure.sh
set -u
OUT=gpurun_out; mkdir -p $OUT
python -m pytest tests/test_gpu_tgrid_tiles.py tests/test_gpu_hashgrid.py tests/test_gpu_nerfplayer_trainer.py tests/test_gpu_nerfplayer_full_trainer.py \
  tests/test_gpu_default_path.py tests/test_gpu_mlp_rows.py tests/test_gpu_headline_step.py tests/test_gpu_kplanes.py -x -q 2>&1 | tail -8
{
echo "# tools/bench_nerfplayer.py --fused --stadium [flags]  (config 4, stadium scene's camera rays)"
for a in "" "--tiled --late-bin" "--tiled" "" "--tiled --late-bin" "--tiled"; do echo -n "[$a] "; python tools/bench_nerfplayer.py --fused --stadium $a 2>/dev/null | tail -1; done
echo "# tools/bench_nerfplayer_full.py [flags]  (full NeRFPlayer preset, bf16 operands, asynchronous sweeps)"
for a in "" "--tiled --late-bin" "--tiled" "--tiled --tiled-hash" "" "--tiled" "--tiled --tiled-hash"; do echo -n "[$a] "; python tools/bench_nerfplayer_full.py $a 2>/dev/null | tail -1; done
} > $OUT/r06_ab_lines.txt 2>&1
cat $OUT/r06_ab_lines.txt
bash tools/collect_config4_profiles.sh r06 2>&1 | tail -4
bash tools/collect_nerfplayer_full_profiles.sh r06 2>&1 | tail -3

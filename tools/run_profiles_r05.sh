#!/bin/bash
# Round-5 profile set (run on the GPU box from the repo root); every artefact lands in gpurun_out/r05_* and is copied to profiles/ by hand.
set -u
bash tools/collect_profiles.sh r05 > gpurun_out/r05_collect_profiles.log 2>&1 || echo "collect_profiles failed" >&2
bash tools/collect_mfma_pmc.sh r05 > gpurun_out/r05_collect_mfma.log 2>&1 || echo "collect_mfma_pmc failed" >&2
bash tools/timeline.sh r05 > gpurun_out/r05_timeline.log 2>&1 || echo "timeline failed" >&2
bash tools/collect_config4_profiles.sh r05 > gpurun_out/r05_collect_config4.log 2>&1 || echo "collect_config4 failed" >&2
bash tools/collect_nerfplayer_full_profiles.sh r05 > gpurun_out/r05_collect_npf.log 2>&1 || echo "collect_nerfplayer_full failed" >&2
ls -la gpurun_out | grep r05_ | tail -40

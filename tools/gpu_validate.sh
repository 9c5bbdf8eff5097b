#!/bin/bash
# Validation call of a round = one tools/gpu_steps.sh call: the library loads, the whole -m gpu suite, per-shape fabric traffic of
# THIS build (bench.py reads it, digest-keyed), the default bench line with both baseline legs, the configs[2] and 448x768 lines,
# kernel trace of a 10 + 10-step clip, SQ counters, the training step, the multi-rank branches on one rank.
#   gpurun --timeout 2700 -- 'bash tools/gpu_validate.sh r05v'      then      bash tools/collect_profiles.sh r05v r05
exec bash "$(dirname "$0")/gpu_steps.sh" "${1:-r05v}" lib pytest pmcshape bench benchcfg3 bench448 trace pmcsq train dist1 smoke

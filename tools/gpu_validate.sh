#!/bin/bash
# Validation call of a round = one tools/gpu_steps.sh call: the library loads, the whole -m gpu suite, per-shape fabric traffic of
# THIS build at one clip per step (the headline) AND at four (the throughput leg) — bench.py reads both entries, digest-keyed —, the
# default bench line with both legs and both baselines, the configs[2] and 448x768 lines, kernel traces of a 10 + 10-step clip at both
# batch sizes, SQ counters, the training step, the multi-rank branches on one rank.
#   gpurun --timeout 3300 -- 'bash tools/gpu_validate.sh r06v'      then      bash tools/collect_profiles.sh r06v r06
exec bash "$(dirname "$0")/gpu_steps.sh" "${1:-r06v}" lib pytest pmcshape:1 pmcshape:4 bench benchcfg3 bench448 trace trace:4 pmcsq train dist1 smoke

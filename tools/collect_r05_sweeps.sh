#!/bin/bash
# Round 5: the sweeps and probes the reference-start schedule was chosen on, in the order they were run (each a GPU call of 1-3 minutes;
# outputs under gpurun_out/r05?/, copied into profiles/r05_reference_start_sweep_*.txt by tools/refresh_r05.py).  The variant tables live in
# tools/verdict_sweep.py (VARIANTS) as of the commit that ran them; the last table is the one in the tree.
#   1  second attempts on the grid (round 4's first attempt): pose-only phases on a fourth level rescue 8 of 14, the other IRLS epsilon 14 of 14
#      python tools/verdict_sweep.py --npz gpurun_out/r05a/verdict_sweep.npz
#   2  the same on ragged masks: 6 % first-attempt failures, half rescued; silent local minima -> the batch-relative cost test
#      python tools/verdict_sweep.py --shape blobs --starts 3072 --alone ""
#   3  depth damping (SP_PHASE_DEPTH_DAMP): strengths 4 / 8 / 16, caps 8 / 12 / 16 / 25, with / without the undamped phase, 384 and 768 slots
#      python tools/verdict_sweep.py [--shape blobs --starts 3072] [--slots 768] --alone "" --variants ...
#   4  damping 16 / 31 and damped second attempts over 12288 ragged starts
#      python tools/verdict_sweep.py --shape blobs --starts 12288 --alone "" --variants d16c12,d16c16,d31c12,d31c16,...
#   5  the two g20y starts alone (64-point spans) and replicated, the starts that fail twice under exotic variants
#      python tools/alone_probe.py all;  python tools/hard_ragged_probe.py 2437 9847 8479
#   6  kernel experiments: one reciprocal for the three IRLS weights; occupancy 3 / 2 through an LDS allocation
#      tools/build_variant.sh ... ;  bash tools/ab_kbench.sh "base onercp" 1 --granule 64 ;  bash tools/ab_kbench.sh "base occ3 occ2" 1 --granule 64
echo "see the comments; the final record of the shipped schedule is made by tools/collect_r05.sh"

# the largest sweep of the round (profiles/r05_reference_start_sweep_6_49152_starts.txt): shipped schedule only, 49152 starts each
#   python tools/verdict_sweep.py --starts 49152 --slots 768 --alone "" --variants shipped
#   python tools/verdict_sweep.py --shape blobs --starts 49152 --alone "" --variants shipped

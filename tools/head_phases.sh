#!/bin/bash
# Phase profile of head_bwd_pc_kernel's producer waves (cycles per 16-pixel group) + the ablation timings.  Needs the profiling
# build of the library:  hipcc ... -DPOPCORN_HEAD_PROF  -> ab/libpopcorn_prof.so  (head backward)
: "${GRAFT_REPO_ROOT:?run on the GPU box through gpurun}"
cd "$GRAFT_REPO_ROOT" || exit 1
python tools/ablate_head.py
[ -f ab/libpopcorn_prof.so ] && POPCORN_HIP_LIB=ab/libpopcorn_prof.so POPCORN_HEAD_PROF=1 python tools/ablate_head.py --phases

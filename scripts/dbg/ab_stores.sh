#!/bin/bash
# What do the workspace stores of the generated routines cost? (dev probe, run on the GPU box through gpurun; ~10 minutes: the generator runs three times)
# base -> a throw-away build without the stores (wrong results, timing only) -> base again, so that the tree the box holds ends as it began.
cd "$(dirname "$0")/../.."
bash scripts/dbg/ab_gen.sh "base" "no_workspace_stores MBLS_GEN_TIMING_NO_STORES=1" "base_again"

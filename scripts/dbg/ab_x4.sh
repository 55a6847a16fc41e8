#!/bin/bash
# What would a workspace of 16-byte vectors per lane buy (a quarter of the memory instructions, the same bytes)? A throw-away build of the three big routines with
# dwordx4 accesses (wrong results, timing only), beside the base. Dev probe, run on the GPU box through gpurun.
cd "$(dirname "$0")/../.."
bash scripts/dbg/ab_gen.sh "base" "x4 MBLS_GEN_TIMING_X4=1" "x4_no_lds MBLS_GEN_TIMING_X4=1 MBLS_GEN_TIMING_NO_LDS=1" "base_again"

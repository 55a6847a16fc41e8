#!/bin/bash
# What does ISSUING the memory instructions cost a lone wave (not waiting for them: scripts/dbg/ab_waits.sh)? Throw-away builds of the three big routines without their
# LDS instructions / workspace loads / workspace stores (wrong results, timing only), then the base again. Dev probe, run on the GPU box through gpurun.
cd "$(dirname "$0")/../.."
bash scripts/dbg/ab_gen.sh "base" "no_lds MBLS_GEN_TIMING_NO_LDS=1" "no_workspace_loads MBLS_GEN_TIMING_NO_GLOADS=1" "no_workspace_stores MBLS_GEN_TIMING_NO_STORES=1" \
    "none_of_them MBLS_GEN_TIMING_NO_LDS=1 MBLS_GEN_TIMING_NO_GLOADS=1 MBLS_GEN_TIMING_NO_STORES=1 MBLS_GEN_TIMING_NO_VMWAIT=1 MBLS_GEN_TIMING_NO_LGKMWAIT=1" "base_again"

#!/bin/bash
# What do the waits for workspace loads and for LDS reads cost a lone wave? (dev probe, run on the GPU box through gpurun: the generator runs five times)
# base -> throw-away builds of the three big routines without their s_waitcnt vmcnt / lgkmcnt (wrong results, timing only) -> base again.
cd "$(dirname "$0")/../.."
bash scripts/dbg/ab_gen.sh "base" "no_vm_waits MBLS_GEN_TIMING_NO_VMWAIT=1" "no_lds_waits MBLS_GEN_TIMING_NO_LGKMWAIT=1" "neither MBLS_GEN_TIMING_NO_VMWAIT=1 MBLS_GEN_TIMING_NO_LGKMWAIT=1" "base_again"

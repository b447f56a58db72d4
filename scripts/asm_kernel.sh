#!/bin/bash
# Device assembly of one kernel of evg_sched.hip (default: the lean planner) -> /tmp/<tag>.s; prints size figures.
# usage: scripts/asm_kernel.sh <tag> [mangled-name-substring]
R=$(cd "$(dirname "$0")/.." && pwd)
TAG=${1:-k}
SYM=${2:-_ZN3evg14k_plan_distrosILb0ELb0EEEvNS_8PlanArgsE}
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-atomic-optimizer-strategy=None \
  --cuda-device-only -S $R/evergreen_amd/csrc/evg_sched.hip -o /tmp/$TAG-all.s 2>/dev/null
awk -v s="^$SYM:" '$0 ~ s {f=1} f{print} /s_endpgm/{if(f)exit}' /tmp/$TAG-all.s > /tmp/$TAG.s
echo "lines $(wc -l < /tmp/$TAG.s)  VALU $(grep -c '^\s*v_' /tmp/$TAG.s)  SALU $(grep -c '^\s*s_' /tmp/$TAG.s)  LDS $(grep -c '^\s*ds_' /tmp/$TAG.s)"
grep -A40 "^\s*.amdhsa_kernel $SYM" /tmp/$TAG-all.s | grep -E "next_free_vgpr|next_free_sgpr|group_segment|private_segment_fixed" 
grep -E "^; (SGPRSpill|VGPRSpill|ScratchSize|Occupancy|NumVgprs|NumSgprs|sgpr_spill|vgpr_spill)" /tmp/$TAG-all.s | head -0

#!/bin/bash
set -uo pipefail
cd "$(dirname "$0")/.."
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM GRBM_GUI_ACTIVE --output-format csv -d $O/insts -- python3 tools/prof_step.py --batch 65536 > $O/insts.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv -d $O/active -- python3 tools/prof_step.py --batch 65536 > $O/active.log 2>&1
(python tools/pmc_insts.py $O/insts tail_fwd_bwd; python tools/pmc_insts.py $O/active tail_fwd_bwd) > $O/tail_instruction_mix_b65536.txt 2>&1; cat $O/tail_instruction_mix_b65536.txt | head -60
rm -rf $O/insts $O/active
rm -f st-dadk_amd/csrc/obj/*.o
STDADK_EXTRA_FLAGS=-DSTDADK_DIAG bash st-dadk_amd/csrc/build.sh > $O/build_diag.log 2>&1 || exit 1
python tools/stamp_tail.py 65536 2>&1 | grep -v amdgpu.ids > $O/stamps_b65536.txt; cat $O/stamps_b65536.txt
python tools/stamp_tail.py 4096 2>&1 | grep -v amdgpu.ids > $O/stamps_b4096.txt; cat $O/stamps_b4096.txt

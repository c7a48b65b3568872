#!/bin/bash
# dynamic instruction mix and wait counters of the decoder's kernels (two counter passes over the full-size decode)
cd $GRAFT_REPO_ROOT
bash tools/pmc_decode.sh a SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS &&
bash tools/pmc_decode.sh b SQ_THREAD_CYCLES_VALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_WAIT_ANY SQ_ACTIVE_INST_SCA &&
python3 tools/pmc_summary.py gpurun_out/pmc_dec_a gpurun_out/pmc_dec_b 2>&1 | grep -E "^==|k_par|k_idct|k_upsample|k_clean" | tee gpurun_out/decode_pmc_r3b.txt

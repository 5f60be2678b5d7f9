#!/bin/bash
# Dev aid (GPU box): kernel trace and two counter passes (issue mix; L1 / L2 requests) of the cross-attention mixin's forward or
# forward + backward.   usage: tools/r3_attn_pmc.sh [--backward] [--bf16]
set -o pipefail
bash tools/prof_cmd.sh attn_sq "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VALU" python3 tools/bench_cross_attn.py "$@" | grep -E "attn|calls" | cut -c1-150 &&
bash tools/prof_cmd.sh attn_tc "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum" python3 tools/bench_cross_attn.py "$@" | grep -E "attn" | grep -v calls | cut -c1-150

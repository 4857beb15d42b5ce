#!/bin/bash
# usage: tools/pmc.sh <outdir> -- <program args...>   : separate rocprofv3 --pmc passes (kernel-trace only)
out=$1; shift; shift
export TMPDIR=/tmp
i=0
for ctrs in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU" \
            "SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
            "FETCH_SIZE GRBM_GUI_ACTIVE" \
            "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $out/pass$i -- "$@" > $out.pass$i.log 2>&1 || { echo "pass $i failed"; tail -5 $out.pass$i.log; exit 1; }
done
echo done

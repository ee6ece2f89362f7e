# SQ stall / LDS / cache counters of single goat_gemm_bf16 launches, round-2 tiles (tile = rows | cols << 16)
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/stall2
mkdir -p $O
rm -f $O/summary.txt
i=0
for shape in "3840 3072 768 0 0 16777408 2" "3840 3072 768 0 0 128 258" "3840 3072 768 0 0 16777472 2" "8192 8192 8192 0 0 16777472 2"; do
  tag=$(echo $shape | tr ' ' '_')
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_VALU" "SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python /root/repo/scripts/gemm_stall_pmc.py $shape > $O/p$i.log 2>&1
    echo "== $tag :: $set" >> $O/summary.txt
    python /root/repo/scripts/pmc_summary.py $O/p$i 60 | grep gemm2 >> $O/summary.txt
    rm -rf $O/p$i
  done
done
cat $O/summary.txt

cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/stall2
mkdir -p $O
i=0
for shape in "3072 768 3840 1 1 128 2" "3840 768 3072 0 1 128 3"; do
  tag=$(echo $shape | tr ' ' '_')
  for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_WAIT_INST_LDS" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_INSTS_LDS"; do
    i=$((i+1))
    timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python /root/repo/scripts/gemm_stall_pmc.py $shape > $O/p$i.log 2>&1
    echo "== $tag :: $set" >> $O/summary.txt
    python /root/repo/scripts/pmc_summary.py $O/p$i 60 | grep gemm2 >> $O/summary.txt
    rm -rf $O/p$i
  done
done
cat $O/summary.txt

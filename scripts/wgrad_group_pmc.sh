# cache / LDS / MFMA counters of goat_wgrad_grouped on the text group (16 problems, 3840 rows): scripts/wgrad_group_bench.py under
# rocprofv3 --pmc, one counter set per pass.   bash scripts/wgrad_group_pmc.sh [tile,stages ...]
cd /tmp && export TMPDIR=/tmp
O=/root/repo/gpurun_out/wgpmc
mkdir -p $O
rm -f $O/summary.txt
i=0
for cfg in "${@:-256,3}"; do
  for set in "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS"; do
    i=$((i+1))
    WG_GROUP=0 WG_CFG=$cfg timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $O/p$i -- python /root/repo/scripts/wgrad_group_bench.py > $O/p$i.log 2>&1
    echo "== cfg $cfg :: $set" >> $O/summary.txt
    python /root/repo/scripts/pmc_summary.py $O/p$i 60 | grep gemm2 >> $O/summary.txt
    rm -rf $O/p$i
  done
done
cat $O/summary.txt

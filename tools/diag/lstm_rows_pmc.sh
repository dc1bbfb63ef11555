R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for set in "FETCH_SIZE WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "MemUnitBusy MemUnitStalled" "TA_BUSY_avr TA_BUSY_max" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY" "TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/lr_pmc/p$i -o p -- python3 $R/tools/diag/lstm_rows_time.py > $R/gpurun_out/lr_pmc_$i.log 2>&1
  echo "set $i ($set) rc=$?"
done

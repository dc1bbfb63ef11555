R=$PWD; cd /tmp && export TMPDIR=/tmp
i=0
for set in "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/lf_pmc/p$i -o p -- python3 $R/tools/diag/lstm_forward_time.py > $R/gpurun_out/lf_pmc_$i.log 2>&1
  echo "set $i ($set) rc=$?"
done

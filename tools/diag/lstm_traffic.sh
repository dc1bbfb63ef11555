# Fabric traffic of the recurrent path's three big kernels (request counters of the L2's memory side; separate passes,
# kernel trace only): training forward step, backward through time, weight gradient -- on the recurrent bench's shape.
#   gpurun -- 'bash tools/diag/lstm_traffic.sh'   -> gpurun_out/lstm_traffic.txt
R=$PWD; cd /tmp && export TMPDIR=/tmp
for prog in lstm_forward_time lstm_rows_time; do
  timeout -k 10 200 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv \
      -d $R/gpurun_out/lt_pmc/$prog -o p -- python3 $R/tools/diag/$prog.py > $R/gpurun_out/lt_$prog.log 2>&1
  echo "$prog rc=$?"
done
cd $R && python3 - <<'PY' > gpurun_out/lstm_traffic.txt
import csv, collections, glob
print("kernel | launches | read requests | of them 32 B | write requests | of them 64 B   (per launch; a wide coalesced read is one 128-byte request)")
for f in sorted(glob.glob("gpurun_out/lt_pmc/*/p_counter_collection.csv")):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if "lstm" in k:
            n = len(v["TCC_EA0_RDREQ_sum"])
            m = lambda c: sum(v[c]) / max(len(v[c]), 1)
            print(f"{k} | {n} | {m('TCC_EA0_RDREQ_sum'):.0f} | {m('TCC_EA0_RDREQ_32B_sum'):.0f} | {m('TCC_EA0_WRREQ_sum'):.0f} | {m('TCC_EA0_WRREQ_64B_sum'):.0f}")
PY
cat gpurun_out/lstm_traffic.txt

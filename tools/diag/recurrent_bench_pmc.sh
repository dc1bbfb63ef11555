# Fabric request counters of every kernel of the recurrent bench (kernel trace only), summarised per kernel.
R=$PWD; cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum --kernel-trace --output-format csv \
    -d $R/gpurun_out/rb_pmc -o p -- python3 $R/bench.py --recurrent --num-envs 8192 --horizon 256 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/rb_pmc.log 2>&1
echo rc=$?
cd $R && python3 - <<'PY'
import csv, collections, glob
f = glob.glob("gpurun_out/rb_pmc/*counter_collection.csv")[0]
# One line per (kernel, grid size): a kernel launched at several sizes (the heads: 2^21 training rows, 8 192 bootstrap
# rows) must not be averaged across them -- round 3's summary did, and bench.py then scaled a 1.56 GB "per launch"
# that was 8/11 of the training launch's 2.17 GB (a traffic ratio of 0.73).
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(f)):
    agg[(r["Kernel_Name"].split("(")[0][:60], int(r.get("Grid_Size", 0) or 0))][r["Counter_Name"]].append(float(r["Counter_Value"]))
rows = []
for (k, grid), v in agg.items():
    n = len(v["TCC_EA0_RDREQ_sum"]); m = lambda c: sum(v[c]) / max(len(v[c]), 1)
    rows.append((sum(v["TCC_EA0_RDREQ_sum"]) + sum(v["TCC_EA0_WRREQ_sum"]), k, grid, n, m("TCC_EA0_RDREQ_sum"), m("TCC_EA0_WRREQ_sum")))
for tot, k, grid, n, rd, wr in sorted(rows, reverse=True)[:16]:
    print(f"{k:62s} grid {grid:10d} launches {n:4d}  reads/launch {rd:12.0f} (x128 B = {rd*128/1e9:6.2f} GB)  writes/launch {wr:12.0f} (x64 B = {wr*64/1e9:6.2f} GB)")
PY

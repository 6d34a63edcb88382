for k in 0; do
rm -rf /tmp/kt_s$k; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_s$k -- python3 bench.py --lib build/libdsim_stop$k.so --workload config5 --steps 100 --warmup 10 --no-cpu-baseline --no-also > /dev/null 2>&1
python3 - <<PY
import csv,glob
for f in glob.glob("/tmp/kt_s$k/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "k_dw_query" in r["Name"] and int(r["Calls"])>50: print("stop $k:", r["Calls"], round(float(r["AverageNs"])/1e3,1), "us")
PY
done

# cfg5 (Mixtral-8x7B shapes) through the unmodified reference host: tokens/s with module stats, then rocprofv3 kernel stats of the same run.
# Run on the GPU box; writes gpurun_out/mixtral/
R=/root/repo; O=$R/gpurun_out/mixtral; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
true
M=/tmp/mixtral-8l.bin; python $R/tools/make_ggmm.py --arch mixtral --config mixtral-8x7b --wtype q4_k --max-len 512 --fast --layers 8 --out $M
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
cd $R/oracle/_ref; rm -rf /tmp/mp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/mp -- ./ref_chat $M all 16 80 - $IDS > /tmp/ids.txt 2> /tmp/mp.err
cp $(find /tmp/mp -name "*kernel_stats.csv" | head -1) $O/kernel_stats.csv
python - <<PY
import csv,glob
f=glob.glob("/tmp/mp/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("kernels",calls,"GPU busy ms",round(tot/1e6,1))
for r in rows[:24]: print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,2), r["Percentage"])
PY
grep "^decode" /tmp/mp.err; tail -5 /tmp/mp.err

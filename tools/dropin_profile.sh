# rocprofv3 kernel stats of the UNMODIFIED reference host (oracle/_ref/ref_chat) decoding on our ggml module (Llama-3-8B shapes, Q4_K).
# Run on the GPU box; writes gpurun_out/dropin_kernel_stats.csv
cd /tmp && export TMPDIR=/tmp
M=/tmp/llama3-8b-q4k.bin
[ -s $M ] || python /root/repo/tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 512 --fast --out $M > /dev/null
cd /root/repo/oracle/_ref
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
rm -rf /tmp/dp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -- ./ref_chat $M all 16 100 - $IDS > /tmp/ids.txt 2> /tmp/dp.err
mkdir -p /root/repo/gpurun_out
cp $(find /tmp/dp -name "*kernel_stats.csv" | head -1) /root/repo/gpurun_out/dropin_kernel_stats.csv
python - <<PY
import csv,glob
f=glob.glob("/tmp/dp/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("kernels",calls,"GPU busy ms",round(tot/1e6,1))
for r in rows[:10]: print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
PY
grep "^decode" /tmp/dp.err

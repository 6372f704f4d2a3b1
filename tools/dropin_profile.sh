cd /tmp && export TMPDIR=/tmp
python /root/repo/tools/make_ggmm.py --config llama3-8b --wtype q4_k --max-len 512 --fast --out /tmp/m.bin > /dev/null
cd /root/repo/oracle/_ref
IDS="1 5 9 200 31 7 11 300 2 77 123 4567 89 1000 2000 3000"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -- ./ref_chat /tmp/m.bin all 16 100 /tmp/l.bin $IDS > /tmp/ids.txt 2> /tmp/dp.err
python - <<PY
import csv,glob
f=glob.glob("/tmp/dp/**/*kernel_stats.csv",recursive=True)[0]
rows=list(csv.DictReader(open(f)))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("kernels",calls,"GPU busy ms",tot/1e6)
for r in rows[:12]: print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e3,1), r["Percentage"])
t=glob.glob("/tmp/dp/**/*kernel_trace.csv",recursive=True)[0]
tr=sorted(csv.DictReader(open(t)), key=lambda r:int(r["Start_Timestamp"]))
# last 60 tokens: wall between first and last kernel of the tail
tail=tr[-60*800:]
print("tail kernels",len(tail),"wall ms",(int(tail[-1]["End_Timestamp"])-int(tail[0]["Start_Timestamp"]))/1e6,"busy ms",sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in tail)/1e6)
PY

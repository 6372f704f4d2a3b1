python /root/repo/tools/make_ggmm.py --config tiny --wtype q4_k --max-len 64 --out /tmp/tiny.bin > /dev/null
cd /root/repo/oracle/_ref
CLLM_HIP_TRACE=1 ./ref_chat /tmp/tiny.bin all 4 1 /tmp/l.bin 1 5 9 2> /tmp/tr.err > /dev/null
grep -n "graph_compute:" /tmp/tr.err | head
awk '/graph_compute:/{c++} c==2' /tmp/tr.err | head -48

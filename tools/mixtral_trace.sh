# node trace + fusion plan of one Mixtral decode graph on the module (tiny model); run on the GPU box
python /root/repo/tools/make_ggmm.py --config tiny --wtype q4_k --max-len 64 --arch mixtral --out /tmp/mx.bin > /dev/null
cd /root/repo/oracle/_ref
CLLM_HIP_TRACE=1 ./ref_chat /tmp/mx.bin all 4 2 /tmp/l.bin 1 5 9 2> /tmp/tr.err > /dev/null
awk '/graph_compute:/{c++} c==4' /tmp/tr.err | cut -c1-230

#!/bin/bash
# round 4, call 37: where the runtime keeps kernel arguments (HIP_FORCE_DEV_KERNARG = 0 / 1 / unset) against the decode step (161 graph kernel nodes per token)
O=gpurun_out/r4_37; mkdir -p $O
B="python bench.py --no-cpu-baseline --no-pmc --no-kernels"
for v in unset 0 1; do
  if [ $v = unset ]; then unset HIP_FORCE_DEV_KERNARG; else export HIP_FORCE_DEV_KERNARG=$v; fi
  for rep in 1 2; do
    $B --steps 20 --warmup 5 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v steps20  %.1f tok/s' % d['value'])" | tee -a $O/summary.txt
  done
  $B --steps 256 --warmup 16 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('HIP_FORCE_DEV_KERNARG=$v steps256 %.1f tok/s' % d['value'])" | tee -a $O/summary.txt
done

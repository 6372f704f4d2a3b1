#!/bin/bash
O=gpurun_out/r3z; mkdir -p $O
export CLLM_SKIP_CFG3=1
timeout 1200 python -m pytest tests/test_gpu_dropin.py -q -x -k "every_length_class or long_prompt" --durations=5 2>&1 | tail -12 | tee $O/pytest_dropin.txt
CLLM_PREFILL=fast timeout 900 python -m pytest tests/test_gpu_llama.py tests/test_gpu_ops.py -q -x -k "not exact and not bit_identical_to_the_oracle and not many_columns" 2>&1 | tail -5 | tee $O/pytest_fast_mode.txt

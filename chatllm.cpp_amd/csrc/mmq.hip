// mmq.hip -- quantized mat-mat (prefill GEMM) on int8 MFMA.  (placeholder: filled in below)
#include "common.h"
int launch_mmq(hipStream_t, int, const tview &, const void *, size_t, const tview &, const tview &) { return CLLM_E_UNSUPPORTED; }

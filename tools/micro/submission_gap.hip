// tools/micro/submission_gap.hip -- what the decode-ahead's per-token stream submissions cost on the GPU's timeline (round 6): a captured graph of 163 short kernels (the step),
// replayed back to back, against the same graph with the module's in-between work submitted separately: an H2D copy of 528 bytes from page-locked memory, a plain kernel launch,
// an event record.  hipcc --offload-arch=gfx950 -O2 -o /tmp/submission_gap tools/micro/submission_gap.hip && /tmp/submission_gap
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_step(float * p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] = p[i] * 1.0001f + 1.0f; }
__global__ void k_prep(float * p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 2.0f; }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float * d; const int n = 256 * 256; CK(hipMalloc(&d, n * 4)); CK(hipMemset(d, 0, n * 4));
    void * tab_d; CK(hipMalloc(&tab_d, 4096)); void * tab_h; CK(hipHostMalloc(&tab_h, 4096));
    hipEvent_t ev, e0, e1; CK(hipEventCreate(&ev)); CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipGraph_t g; hipGraphExec_t step, step_prep;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < 163; i++) hipLaunchKernelGGL(k_step, dim3(256), dim3(256), 0, st, d, n);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&step, g, nullptr, nullptr, 0));
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    hipLaunchKernelGGL(k_prep, dim3(256), dim3(256), 0, st, d, n);
    for (int i = 0; i < 163; i++) hipLaunchKernelGGL(k_step, dim3(256), dim3(256), 0, st, d, n);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&step_prep, g, nullptr, nullptr, 0));
    const int iters = 300;
    const char * names[] = { "graph only", "graph + kernel", "graph + kernel + event record", "graph + H2D 528 B + kernel + event record (the module today)", "ONE graph with the kernel inside",
                             "graph + kernel + event record, host waits for the event every iteration (the module's synchronize)" };
    for (int rep = 0; rep < 2; rep++)
    for (int v = 0; v < 6; v++) {
        for (int pass = 0; pass < 2; pass++) {
            if (pass) CK(hipEventRecord(e0, st));
            for (int i = 0; i < (pass ? iters : 20); i++) {
                if (v == 3) CK(hipMemcpyAsync(tab_d, tab_h, 528, hipMemcpyHostToDevice, st));
                if (v >= 1 && v != 4) hipLaunchKernelGGL(k_prep, dim3(256), dim3(256), 0, st, d, n);
                if (v == 2 || v == 3 || v == 5) CK(hipEventRecord(ev, st));
                CK(hipGraphLaunch(v == 4 ? step_prep : step, st));
                if (v == 5) CK(hipEventSynchronize(ev));
            }
        }
        CK(hipEventRecord(e1, st)); CK(hipEventSynchronize(e1));
        float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-105s %8.1f us per iteration\n", names[v], ms * 1e3f / iters);
    }
    return 0;
}

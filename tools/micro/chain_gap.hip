// tools/micro/chain_gap.hip -- the module's chained decode-ahead in miniature (round 6): per token the stream sees [H2D 528 B][prep kernel][event record][graph of 163 kernels],
// queued while the previous token's graph still runs; the host waits for the previous token's event, reads 513 KB back on the null stream, works ~0.5 ms, queues the next token.
// Where does the GPU idle?  Token period on the GPU's clock (events around each graph) with pieces of the host side switched off.
// hipcc --offload-arch=gfx950 -O2 -o tools/micro/bin/chain_gap tools/micro/chain_gap.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
#include <chrono>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void k_step(float * p, int n) { for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = p[i] * 1.0001f + 1.0f; }
__global__ void k_prep(float * p, int n) { const int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 2.0f; }
static void spin_us(int us) { const auto t0 = std::chrono::steady_clock::now(); while (std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < us) {} }
int main() {
    hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float * d; const int n = 4 << 20; CK(hipMalloc(&d, (size_t) n * 4)); CK(hipMemset(d, 0, (size_t) n * 4));
    void * tab_d; CK(hipMalloc(&tab_d, 4096)); void * tab_h; CK(hipHostMalloc(&tab_h, 4096));
    void * snap; CK(hipMalloc(&snap, 1 << 20)); void * stage; CK(hipHostMalloc(&stage, 1 << 20));
    hipGraph_t g; hipGraphExec_t step;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed));
    for (int i = 0; i < 163; i++) hipLaunchKernelGGL(k_step, dim3(1024), dim3(256), 0, st, d, n);
    CK(hipStreamEndCapture(st, &g)); CK(hipGraphInstantiate(&step, g, nullptr, nullptr, 0));
    const int N = 120;
    static hipEvent_t ev[N + 1], g0[N + 1], g1[N + 1];
    for (int i = 0; i <= N; i++) { CK(hipEventCreate(&ev[i])); CK(hipEventCreate(&g0[i])); CK(hipEventCreate(&g1[i])); }
    struct { const char * name; bool h2d, d2h, work, evsync; } V[] = {
        { "everything (the module's chained loop)", true, true, true, true },
        { "no H2D table copy", false, true, true, true },
        { "no D2H read-back on the null stream", true, false, true, true },
        { "no host work between the tokens", true, true, false, true },
        { "no H2D, no D2H", false, false, true, true },
        { "host never waits (everything queued up front)", true, false, false, false },
    };
    for (int v = 0; v < 6; v++) {
        auto enqueue = [&](int i) {
            if (V[v].h2d) (void) hipMemcpyAsync(tab_d, tab_h, 528, hipMemcpyHostToDevice, st);
            hipLaunchKernelGGL(k_prep, dim3(256), dim3(256), 0, st, d, 65536);
            (void) hipEventRecord(ev[i], st);
            (void) hipEventRecord(g0[i], st);
            (void) hipGraphLaunch(step, st);
            (void) hipEventRecord(g1[i], st);
        };
        enqueue(0);
        for (int i = 1; i <= N; i++) {
            enqueue(i);
            if (V[v].evsync) CK(hipEventSynchronize(ev[i - 1]));
            if (V[v].d2h) { CK(hipMemcpyAsync(stage, snap, 513024, hipMemcpyDeviceToHost, nullptr)); CK(hipStreamSynchronize(nullptr)); }
            if (V[v].work) spin_us(500);
        }
        CK(hipStreamSynchronize(st));
        double graph = 0, period = 0; int cnt = 0;
        for (int i = 20; i < N; i++) { float a = 0, b = 0; CK(hipEventElapsedTime(&a, g0[i], g1[i])); CK(hipEventElapsedTime(&b, g0[i], g0[i + 1])); graph += a; period += b; cnt++; }
        printf("%-50s token period %8.1f us, graph %8.1f us, between two graphs %6.1f us\n", V[v].name, period / cnt * 1e3, graph / cnt * 1e3, (period - graph) / cnt * 1e3);
    }
    return 0;
}

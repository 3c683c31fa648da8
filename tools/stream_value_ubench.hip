// round 6 micro-benchmark: a host <-> GPU ping-pong of N rounds, two ways
//   (a) launch per round + hipStreamQuery polling (what the k-means++ rounds do today)
//   (b) everything enqueued up front: [kernel i ; hipStreamWriteValue32(done, i + 1) ; hipStreamWaitValue32(go, i + 2)] -- the host answers
//       a round by two plain stores (the kernel's argument in pinned memory, then `go`)
// hipcc --offload-arch=gfx950 -O2 tools/stream_value_ubench.hip -o tools/_stream_value_ubench && timeout 60 tools/_stream_value_ubench
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <atomic>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
__global__ void work(const volatile unsigned *arg, float *data, unsigned n, unsigned *out) {
    const unsigned i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) data[i] = data[i] * 0.999f + 1.0f;
    if (i == 0) *out = *arg;                       // the host's answer to the previous round: ONE read of pinned host memory
}
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const int N = argc > 1 ? std::atoi(argv[1]) : 1000;
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    float *d; const unsigned n = 50000; CK(hipMalloc(&d, n * sizeof(float))); CK(hipMemset(d, 0, n * sizeof(float)));
    unsigned *h_arg, *d_out; CK(hipHostMalloc(&h_arg, 4096, hipHostMallocMapped)); CK(hipMalloc(&d_out, 4));
    h_arg[0] = 1;
    // (a)
    for (int rep = 0; rep < 2; ++rep) {
        const double t0 = now();
        for (int i = 0; i < N; ++i) {
            h_arg[0] = (unsigned)i;
            hipLaunchKernelGGL(work, dim3((n + 255) / 256), dim3(256), 0, s, h_arg, d, n, d_out);
            while (hipStreamQuery(s) == hipErrorNotReady) { }
        }
        std::printf("(a) launch + hipStreamQuery per round: %.2f us / round\n", (now() - t0) / N * 1e6);
    }
    // (b)
    unsigned *sig = nullptr;
    hipError_t e = hipExtMallocWithFlags((void **)&sig, 64, hipMallocSignalMemory);
    std::printf("hipMallocSignalMemory: %s\n", hipGetErrorString(e));
    unsigned *done_p, *go_p;
    if (e == hipSuccess) { done_p = sig; go_p = sig + 2; }
    else { (void)hipGetLastError(); done_p = h_arg + 64; go_p = h_arg + 128; }
    volatile unsigned *done = done_p, *go = go_p;
    *done = 0; *go = 0;
    CK(hipStreamSynchronize(s));
    const int M = N < 400 ? N : 400;               // (commands enqueued up front: keep the queue modest)
    for (int rep = 0; rep < 2; ++rep) {
        *done = 0; *go = 0; std::atomic_thread_fence(std::memory_order_seq_cst);
        const unsigned base = 1000u * (unsigned)(rep + 1);
        for (int i = 0; i < M; ++i) {
            hipLaunchKernelGGL(work, dim3((n + 255) / 256), dim3(256), 0, s, h_arg, d, n, d_out);
            e = hipStreamWriteValue32(s, (void *)done_p, base + (unsigned)i + 1u, 0);
            if (e != hipSuccess) { std::printf("hipStreamWriteValue32: %s\n", hipGetErrorString(e)); return 0; }
            e = hipStreamWaitValue32(s, (void *)go_p, base + (unsigned)i + 1u, hipStreamWaitValueGte, 0xFFFFFFFFu);
            if (e != hipSuccess) { std::printf("hipStreamWaitValue32: %s\n", hipGetErrorString(e)); return 0; }
        }
        const double t0 = now();
        bool ok = true;
        for (int i = 0; i < M && ok; ++i) {
            const double tw = now();
            while (*done != base + (unsigned)i + 1u) { if (now() - tw > 5.0) { std::printf("(b) round %d: no completion value after 5 s -- giving up\n", i); ok = false; break; } }
            h_arg[0] = (unsigned)i;                                    // the "pick"
            std::atomic_thread_fence(std::memory_order_seq_cst);
            *go = base + (unsigned)i + 1u;
        }
        if (!ok) { *go = 0x7FFFFFFFu; (void)hipStreamSynchronize(s); return 0; }
        CK(hipStreamSynchronize(s));
        std::printf("(b) pre-enqueued kernel + write / wait values: %.2f us / round\n", (now() - t0) / M * 1e6);
    }
    return 0;
}

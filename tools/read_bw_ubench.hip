// round 6: what a READ-ONLY stream reaches on this part (the guide's 6.29 TB/s is a float4 COPY: reads + writes)
// hipcc --offload-arch=gfx950 -O3 tools/read_bw_ubench.hip -o tools/_read_bw_ubench && tools/_read_bw_ubench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(2); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
template <int U, int AUX>
__global__ __launch_bounds__(256) void rd(const f4 *__restrict__ p, size_t n4, float *out) {
    const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc(const_cast<f4 *>(p), 0, 0x7FFFFFFF, 0x00020000);
    (void)r;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t i = (size_t)blockIdx.x * 256 * U + threadIdx.x; i + 256 * (U - 1) < n4; i += stride) {
        f4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = AUX ? __builtin_nontemporal_load(p + i + 256 * u) : p[i + 256 * u];
#pragma unroll
        for (int u = 0; u < U; ++u) acc += v[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) *out = acc.x;
}
template <int U, int AUX>
static void run(const f4 *d, size_t n4, float *out, int blocks, const char *name) {
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((rd<U, AUX>), dim3(blocks), dim3(256), 0, 0, d, n4, out);
    CK(hipEventRecord(a));
    const int reps = 10;
    for (int w = 0; w < reps; ++w) hipLaunchKernelGGL((rd<U, AUX>), dim3(blocks), dim3(256), 0, 0, d, n4, out);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    std::printf("%-28s blocks %6d: %.2f TB/s\n", name, blocks, (double)n4 * 16 * reps / (ms * 1e-3) / 1e12);
}
__global__ __launch_bounds__(256) void cp(const f4 *__restrict__ p, f4 *__restrict__ q, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) q[i] = p[i];
}
int main() {
    const size_t bytes = 16ull << 30, n4 = bytes / 16;
    f4 *d, *e; float *out; CK(hipMalloc(&d, bytes)); CK(hipMalloc(&e, bytes / 2)); CK(hipMalloc(&out, 4));
    CK(hipMemset(d, 1, bytes));
    for (int blocks : {2048, 8192, 32768}) {
        run<4, 0>(d, n4, out, blocks, "read float4 x4 in flight");
        run<8, 0>(d, n4, out, blocks, "read float4 x8 in flight");
        run<8, 1>(d, n4, out, blocks, "read float4 x8, nontemporal");
    }
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    hipLaunchKernelGGL(cp, dim3(16384), dim3(256), 0, 0, d, e, n4 / 2);
    CK(hipEventRecord(a));
    for (int w = 0; w < 10; ++w) hipLaunchKernelGGL(cp, dim3(16384), dim3(256), 0, 0, d, e, n4 / 2);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms = 0; CK(hipEventElapsedTime(&ms, a, b));
    std::printf("float4 copy (read + write bytes): %.2f TB/s\n", (double)(n4 / 2) * 32 * 10 / (ms * 1e-3) / 1e12);
    return 0;
}

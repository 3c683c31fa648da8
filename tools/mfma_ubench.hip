// f32 MFMA throughput microbenchmark: independent accumulator chains of v_mfma_f32_16x16x4_f32 and
// v_mfma_f32_32x32x2_f32, 1..8 waves per SIMD.  Prints achieved TFLOP/s and cycles per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));

template <int CHAINS>
__global__ __launch_bounds__(256) void k16(float *out, const float *in, int iters) {
    f4 acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    const float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int CHAINS>
__global__ __launch_bounds__(256) void k32(float *out, const float *in, int iters) {
    f16v acc[CHAINS];
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) acc[c][i] = 0.f;
    const float a = in[threadIdx.x], b = in[threadIdx.x + 256];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < CHAINS; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CHAINS; ++c) for (int i = 0; i < 16; ++i) s += acc[c][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <class K> void run(const char *name, K kern, int chains, double flops_per_inst, int blocks_per_cu, float *out, float *in) {
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, in, 10);
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * 4 * chains;          // wave instructions
    printf("%-34s waves/SIMD %d  %8.3f ms  %7.2f TFLOP/s  %.1f SIMD-cycles/inst at 2.4 GHz\n", name, blocks_per_cu, ms,
           insts * flops_per_inst / ms / 1e9, ms * 1e-3 * 2.4e9 * 1024 / insts);
}

int main() {
    float *out, *in; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&in, 4096);
    hipMemset(in, 0, 4096);
    for (int w : {1, 2, 4}) {
        run("16x16x4 f32, 1 chain", k16<1>, 1, 2048.0, w, out, in);
        run("16x16x4 f32, 4 chains", k16<4>, 4, 2048.0, w, out, in);
        run("16x16x4 f32, 16 chains", k16<16>, 16, 2048.0, w, out, in);
        run("32x32x2 f32, 1 chain", k32<1>, 1, 4096.0, w, out, in);
        run("32x32x2 f32, 4 chains", k32<4>, 4, 4096.0, w, out, in);
    }
    return 0;
}

// VALU throughput microbenchmark: scalar vs packed f32 add/mul chains, SGPR operand or not.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, const float *in, int iters) {
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[threadIdx.x + i * 256]; b[i] = in[threadIdx.x + 2048 + i * 256]; }
    const float s0 = in[4096], s1 = in[4097];   // wave-uniform operands
    v2f pa[4], pb[4];
    for (int i = 0; i < 4; ++i) { pa[i] = (v2f){a[2*i], a[2*i+1]}; pb[i] = (v2f){b[2*i], b[2*i+1]}; }
    const v2f ps = (v2f){s0, s1};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            if (MODE == 0) {        // 8 independent scalar chains: sub, mul, add (3 ops)
#pragma unroll
                for (int i = 0; i < 8; ++i) { float d = s0 - a[i]; d = d * d; a[i] = b[i] + d; }
            } else if (MODE == 1) { // 4 independent packed chains: pk_sub, pk_mul, pk_add
#pragma unroll
                for (int i = 0; i < 4; ++i) { v2f d = ps - pa[i]; d = d * d; pa[i] = pb[i] + d; }
            } else if (MODE == 2) { // scalar, VGPR-only operands
#pragma unroll
                for (int i = 0; i < 8; ++i) { float d = b[(i+1)&7] - a[i]; d = d * d; a[i] = b[i] + d; }
            } else {                // packed, VGPR-only
#pragma unroll
                for (int i = 0; i < 4; ++i) { v2f d = pb[(i+1)&3] - pa[i]; d = d * d; pa[i] = pb[i] + d; }
            }
        }
    }
    float acc = 0;
    for (int i = 0; i < 8; ++i) acc += b[i] + a[i];
    for (int i = 0; i < 4; ++i) acc += pb[i].x + pb[i].y + pa[i].x + pa[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int MODE> void run(const char *name, float *out, float *in) {
    const int iters = 2000, blocks = 256 * 8;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<blocks, 256>>>(out, in, 10);
    hipEventRecord(e0);
    k<MODE><<<blocks, 256>>>(out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // element-ops: per iter per thread: 8 reps * 8 elements * 3 ops
    double ops = (double)blocks * 256 * iters * 8 * 8 * 3;
    double winst = (double)blocks * 4 * iters * 8 * ((MODE & 1) ? 4 * 3 : 8 * 3);   // wave instructions
    double simd_cycles = ms * 1e-3 * 2.4e9 * 1024;  // at nominal 2.4 GHz
    printf("%-28s %8.3f ms  %7.2f Tops/s  %.2f cycles/wave-instr (at 2.4 GHz)\n", name, ms, ops / ms / 1e9, simd_cycles / winst);
}

int main() {
    float *out, *in; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&in, 8192 * 4);
    std::vector<float> h(8192); for (int i = 0; i < 8192; ++i) h[i] = 0.001f * (i % 97);
    hipMemcpy(in, h.data(), 8192 * 4, hipMemcpyHostToDevice);
    run<0>("scalar, SGPR operand", out, in);
    run<1>("packed, SGPR-pair operand", out, in);
    run<2>("scalar, VGPR only", out, in);
    run<3>("packed, VGPR only", out, in);
    return 0;
}

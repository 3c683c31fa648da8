// How much non-MFMA work can ride along with f32 MFMAs?  Each loop iteration issues 16 independent
// v_mfma_f32_16x16x4_f32 plus NV dependent-free VALU ops (and optionally NL 16-byte global loads from
// an L1-hot buffer); 1..4 waves per SIMD.  Prints the MFMA rate achieved.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NV, int NL>
__global__ __launch_bounds__(256) void k(float *out, const float4 *in, int iters) {
    f4 acc[4];
    for (int c = 0; c < 4; ++c) acc[c] = (f4){0.f, 0.f, 0.f, 0.f};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = in[threadIdx.x].x + i;
    float4 a = in[threadIdx.x], b = in[threadIdx.x + 256];
    const float4 *p = in + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
        float4 l[NL > 0 ? NL : 1];
#pragma unroll
        for (int i = 0; i < NL; ++i) l[i] = p[((it + i) & 7) * 256];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b.x, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b.y, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b.z, acc[c], 0, 0, 0);
            acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b.w, acc[c], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < NV; ++i) v[i & 7] = v[i & 7] * 1.0001f + 0.5f;
        if (NL > 0) {
#pragma unroll
            for (int i = 0; i < NL; ++i) { a.x += l[i].x * 1e-30f; b.y += l[i].y * 1e-30f; }
        }
    }
    float s = 0;
    for (int c = 0; c < 4; ++c) s += acc[c][0] + acc[c][1] + acc[c][2] + acc[c][3];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int NL> void run(int blocks_per_cu, float *out, float4 *in) {
    const int iters = 4000, blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NV, NL><<<blocks, 256>>>(out, in, 10);
    hipEventRecord(e0);
    k<NV, NL><<<blocks, 256>>>(out, in, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)blocks * 4 * iters * 16;
    printf("16 MFMA + %2d VALU + %d loads per iteration, %d waves/SIMD: %7.2f TFLOP/s (%.0f %% of 157.3)\n", NV, NL, blocks_per_cu,
           insts * 2048.0 / ms / 1e9, insts * 2048.0 / ms / 1e9 / 157.3 * 100);
}

int main() {
    float *out; float4 *in; hipMalloc(&out, 256 * 8 * 256 * 4); hipMalloc(&in, 64 * 1024);
    hipMemset(in, 0, 64 * 1024);
    for (int w : {1, 2, 3}) {
        run<0, 0>(w, out, in); run<8, 0>(w, out, in); run<16, 0>(w, out, in); run<32, 0>(w, out, in); run<64, 0>(w, out, in);
        run<16, 3>(w, out, in); run<16, 6>(w, out, in);
    }
    return 0;
}

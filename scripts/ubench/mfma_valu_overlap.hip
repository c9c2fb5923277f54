// Microbenchmark: how many independent VALU / DS instructions hide under one
// v_mfma_f32_32x32x16_f16 when a wave is alone on its SIMD (1 wave/SIMD, 4 waves/CU)?
// Prints cycles per MFMA for K = 0..10 filler instructions of several kinds.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int K, int KIND>
__global__ __launch_bounds__(256, 1) void kern(long* out, float seed) {
    __shared__ float lds[4096];
    h8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)(seed + threadIdx.x); b[j] = (_Float16)(seed * 2 + j); }
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float v[12];
    for (int i = 0; i < 12; ++i) v[i] = seed + i + threadIdx.x;
    lds[threadIdx.x] = seed;
    __syncthreads();
    const int ITER = 2000;
    long t0 = clock64();
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[m], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < K; ++k) {
                if (KIND == 0) asm volatile("v_max_f32 %0, 0, %0" : "+v"(v[k]));
                if (KIND == 1) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %0" : "+v"(v[k]));
                if (KIND == 2) asm volatile("v_pk_add_f32 %0, %0, %0" : "+v"(*(double*)&v[(k & ~1) % 12 < 11 ? (k & ~1) % 12 : 0]));
                if (KIND == 3) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
                if (KIND == 4) { float x; asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"((threadIdx.x & 63) * 4)); asm volatile("" :: "v"(x)); }
                if (KIND == 5) asm volatile("v_accvgpr_read_b32 %0, a200" : "=v"(v[k]));
            }
        }
        if (KIND == 4) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    long t1 = clock64();
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    for (int i = 0; i < 12; ++i) s += v[i];
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (s == 1234.5f) out[1] = 1;
}

template <int K, int KIND>
void run(long* d) {
    hipLaunchKernelGGL((kern<K, KIND>), dim3(256), dim3(256), 0, 0, d, 1.0f);
    hipDeviceSynchronize();
    long h;
    hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    printf("kind %d K %2d : %.1f clock64 ticks per MFMA\n", KIND, K, (double)h / (2000.0 * 4));
}

int main() {
    long* d;
    hipMalloc(&d, 64);
#define ROW(KIND) run<0, KIND>(d); run<2, KIND>(d); run<4, KIND>(d); run<6, KIND>(d); run<8, KIND>(d); run<10, KIND>(d);
    ROW(0) ROW(1) ROW(2) ROW(3) ROW(4) ROW(5)
    return 0;
}

// Which SIMD does wave w of a 512-thread (and 256-thread) workgroup land on?  (HW_REG_HW_ID bits 5:4)
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void probe(unsigned* out) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main() {
    unsigned* d; hipMalloc(&d, 4096 * 4);
    for (int bs : {512, 256}) {
        const int nw = bs / 64, nb = 6;
        probe<<<nb, bs>>>(d);
        unsigned h[64]; hipMemcpy(h, d, nb * nw * 4, hipMemcpyDeviceToHost);
        for (int b = 0; b < nb; ++b) {
            printf("bs=%d block %d:", bs, b);
            for (int w = 0; w < nw; ++w) {
                unsigned v = h[b * nw + w];
                printf("  w%d simd%u cu%u se%u wv%u", w, (v >> 4) & 3, (v >> 8) & 15, (v >> 13) & 7, v & 15);
            }
            printf("\n");
        }
    }
    return 0;
}

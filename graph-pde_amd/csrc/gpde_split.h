// Device helpers of the 2-term f16 split shared by the per-edge backward kernels
// (gpde_edge_bwd3.hip, the one-pass backward mode of gpde_fused_f16v6.hip): an operand a*s (s a power of two that puts the
// row's / node's maximum into [2^13, 2^14)) is carried as hi = rtz16(a*s), lo = rn16(a*s - hi); hi*hi + hi*lo + lo*hi on
// v_mfma_f32_32x32x16_f16 has a relative error < 2^-21 per product, fp32 accumulation.
#ifndef GPDE_SPLIT_H
#define GPDE_SPLIT_H
#include <hip/hip_runtime.h>

// y = v * sc; hi = rtz16(y); lo = rn16(y - hi): two values per call, packed halves
__device__ __forceinline__ void gp_split2(float v0, float v1, float sc, unsigned& ph, unsigned& pl) {
    unsigned t0, t1;
    asm("v_mul_f32 %1, %5, %3\n\t"
        "v_mul_f32 %2, %5, %4\n\t"
        "v_cvt_pkrtz_f16_f32 %0, %1, %2"
        : "=&v"(ph), "=&v"(t0), "=&v"(t1) : "v"(v0), "v"(v1), "v"(sc));
    asm("v_fma_mixlo_f16 %0, %1, 1.0, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]\n\t"
        "v_fma_mixhi_f16 %0, %2, 1.0, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
        : "=&v"(pl) : "v"(t0), "v"(t1), "v"(ph));
}
// 2^(13 - floor(log2 m)) and its reciprocal for m in the normal range (else 1, 1): m -> [2^13, 2^14)
__device__ __forceinline__ void gp_pow2_scale(float m, float& sc, float& isc) {
    const int eb = (__float_as_int(m) >> 23) & 0xff;
    const bool ok = eb >= 20 && eb <= 230;
    sc = ok ? __int_as_float((267 - eb) << 23) : 1.f;
    isc = ok ? __int_as_float((eb - 13) << 23) : 1.f;
}
// the value of lane (l ^ 32); every lane of the wave must execute it
__device__ __forceinline__ float gp_other_half(float v) {
    const auto r2 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float((threadIdx.x & 32) ? r2[0] : r2[1]);
}
// the value CTRL lanes down the 16-lane row, 0 beyond it (DPP row_shr)
template <int CTRL> __device__ __forceinline__ float gp_dpp_shr(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
// Sum and maximum of |.| of `v` over the 32 lanes of each half wave by a FIXED DPP tree (row_shr 1, 2, 4, 8, then lane 15 of rows
// 0 / 2 into rows 1 / 3): lanes 31 and 63 end up with their half's totals; the order does not depend on anything -> reproducible.
__device__ __forceinline__ void gp_half_wave_sum_max(float v, float& sum, float& mx) {
    float sv = v, mv = fabsf(v);
    sv += gp_dpp_shr<0x111>(sv); mv = fmaxf(mv, gp_dpp_shr<0x111>(mv));
    sv += gp_dpp_shr<0x112>(sv); mv = fmaxf(mv, gp_dpp_shr<0x112>(mv));
    sv += gp_dpp_shr<0x114>(sv); mv = fmaxf(mv, gp_dpp_shr<0x114>(mv));
    sv += gp_dpp_shr<0x118>(sv); mv = fmaxf(mv, gp_dpp_shr<0x118>(mv));
    sv += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(sv), 0x142, 0xa, 0xf, false));
    mv = fmaxf(mv, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(mv), 0x142, 0xa, 0xf, false)));
    sum = sv; mx = mv;
}
#endif

// Probe: which way do the carry-writing subtractions subtract when src0 carries a DPP modifier?  (lane value = lane id, partner = lane ^ 1)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(int* out, unsigned long long* masks)
{
    uint32_t x = threadIdx.x, t1, t2, t3, t4; unsigned long long m1, m2, m3, m4;
    asm volatile("s_nop 1\n\tv_subrev_co_u32_dpp %0, vcc, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_mov_b64 %1, vcc" : "=&v"(t1), "=s"(m1) : "v"(x) : "vcc");
    asm volatile("s_nop 1\n\tv_sub_co_u32_dpp %0, vcc, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\ts_mov_b64 %1, vcc" : "=&v"(t2), "=s"(m2) : "v"(x) : "vcc");
    uint32_t o = __shfl_xor(x, 1);
    asm volatile("v_subrev_co_u32 %0, vcc, %3, %2\n\ts_mov_b64 %1, vcc" : "=&v"(t3), "=s"(m3) : "v"(x), "v"(o) : "vcc");      // src0 = other, src1 = mine
    asm volatile("v_sub_co_u32 %0, vcc, %3, %2\n\ts_mov_b64 %1, vcc" : "=&v"(t4), "=s"(m4) : "v"(x), "v"(o) : "vcc");
    out[threadIdx.x] = (int)t1; out[64 + threadIdx.x] = (int)t2; out[128 + threadIdx.x] = (int)t3; out[192 + threadIdx.x] = (int)t4;
    if (threadIdx.x == 0) { masks[0] = m1; masks[1] = m2; masks[2] = m3; masks[3] = m4; }
}
int main()
{
    int* d; unsigned long long* m; hipMalloc(&d, 4 * 256); hipMalloc(&m, 32);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, m);
    int h[256]; unsigned long long hm[4]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost); hipMemcpy(hm, m, 32, hipMemcpyDeviceToHost);
    const char* nm[4] = {"v_subrev_co_u32_dpp d, vcc, x(dpp), x", "v_sub_co_u32_dpp    d, vcc, x(dpp), x", "v_subrev_co_u32     d, vcc, other, mine", "v_sub_co_u32        d, vcc, other, mine"};
    for (int i = 0; i < 4; ++i) printf("%s : lane0 %+d lane1 %+d  vcc %016llx\n", nm[i], h[64 * i], h[64 * i + 1], hm[i]);
    return 0;
}

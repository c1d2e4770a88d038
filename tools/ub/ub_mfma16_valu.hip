// reproducer attempt for the float16-pair experiment (xcorr32_mfma16_experiment.hip.txt): do vector results computed in a wave that also
// issues v_mfma_f32_16x16x32_f16 change from launch to launch?  Each wave stages 1 088 pseudo-random samples in LDS (padded rows, as the
// correlation kernel does), forms the 16 window energies per lane (head + core + tail: v_fma_f32 / v_pk_fma_f32 / v_pk_add_f32 chains), and,
// in the MFMA build, runs 36 float16 matrix instructions on float16 copies of the same samples before / after.  The factors 1/sqrt(energy) go
// through the experiment's LDS exchange (owner layout -> accumulator layout -> in-place write-back -> lane order), then factors and accumulators
// go to global memory; the host launches each build three times and counts words that differ between launches and between the builds.
//   hipcc -O3 --offload-arch=gfx950 -o ub_mfma16_valu tools/ub/ub_mfma16_valu.hip && ./ub_mfma16_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#define CK(x) do { hipError_t err_ = (x); if (err_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(err_), __LINE__); exit(1); } } while (0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
constexpr int ROW = 20, SMP = 68 * ROW, HALF = 69 * 24, WAVES = 4;

__device__ __forceinline__ void wave_fence_lds() { __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront"); __builtin_amdgcn_wave_barrier(); }

template <int MODE>       // 0: energies only; 1: matrix phase, then energies; 2: energies, then matrix phase
__global__ __launch_bounds__(64 * WAVES, 3)
void k(const float* __restrict__ y, long long n_items, float* __restrict__ en_out, float* __restrict__ acc_out)
{
    __shared__ __attribute__((aligned(16))) float s_smp[WAVES][SMP];
    __shared__ __attribute__((aligned(16))) _Float16 x_hi[WAVES][HALF];
    __shared__ __attribute__((aligned(16))) _Float16 x_lo[WAVES][HALF];
    const int lane = threadIdx.x & 63, wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int c = lane & 15, q = lane >> 4;
    float* const s = s_smp[wv];
    if (lane < 24) { x_hi[wv][68 * 24 + lane] = (_Float16)0.0f; x_lo[wv][68 * 24 + lane] = (_Float16)0.0f; }
    const unsigned stride = gridDim.x * WAVES;
    f32x4 stage[5];
    auto prefetch = [&](unsigned it) {
        const f32x4* src = reinterpret_cast<const f32x4*>(y + (long long)it * 1024) + lane;
        #pragma unroll
        for (int u = 0; u < 4; ++u) stage[u] = src[64 * u];
        stage[4] = (lane < 16) ? src[256] : f32x4{1.0f, 1.0f, 1.0f, 1.0f};
    };
    unsigned item = blockIdx.x * WAVES + wv;
    if (item >= (unsigned)n_items) return;
    prefetch(item);
    for (; item < (unsigned)n_items; item += stride) {
        #pragma unroll
        for (int u = 0; u < 4; ++u) *reinterpret_cast<f32x4*>(s + 4 * lane + 4 * (lane >> 2) + 320 * u) = stage[u];
        if (lane < 16) *reinterpret_cast<f32x4*>(s + 4 * lane + 4 * (lane >> 2) + 1280) = stage[4];
        #pragma unroll
        for (int u = 0; u < 5; ++u) {
            if (u == 4 && lane >= 16) break;
            const int ph = 4 * lane + 8 * (lane >> 2) + 384 * u;
            f16x4 h4, l4;
            #pragma unroll
            for (int v = 0; v < 4; ++v) {
                const float z = stage[u][v] * 4096.0f;
                const _Float16 zh = (_Float16)z;
                h4[v] = zh; l4[v] = (_Float16)(z - (float)zh);
            }
            *reinterpret_cast<f16x4*>(x_hi[wv] + ph) = h4;
            *reinterpret_cast<f16x4*>(x_lo[wv] + ph) = l4;
        }
        if (item + stride < (unsigned)n_items) prefetch(item + stride);
        wave_fence_lds();
        f32x4 acc[4];
        #pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
        auto matrix = [&]() __attribute__((always_inline)) {
            #pragma unroll
            for (int st = 0; st < 3; ++st) {
                f16x8 ah, al;
                #pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const int kk = 24 * q + 8 * st + j - c;
                    const float tt = (kk >= 0 && kk < 63) ? (float)((kk * 37 + 11) % 64 - 32) * 4.0f + 0.37f : 0.0f;
                    const _Float16 th = (_Float16)tt;
                    ah[j] = th; al[j] = (_Float16)(tt - (float)th);
                }
                const int bo = 24 * c + 24 * q + 8 * st + 8 * ((24 * q + 8 * st) >> 4);
                #pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const f16x8 bh = *reinterpret_cast<const f16x8*>(x_hi[wv] + 384 * t + bo);
                    const f16x8 bl = *reinterpret_cast<const f16x8*>(x_lo[wv] + 384 * t + bo);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc[t], 0, 0, 0);
                }
            }
        };
        if (MODE == 1) matrix();
        const float* w = s + ROW * lane;
        #define LD4(i) (*reinterpret_cast<const f32x4*>(w + 4 * (i) + 4 * ((i) >> 2)))
        float en[16];
        {
            const f32x4 h0 = LD4(0), h1 = LD4(1), h2 = LD4(2), h3 = LD4(3);
            const float hv[16] = {h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3], h2[0], h2[1], h2[2], h2[3], h3[0], h3[1], h3[2], h3[3]};
            en[15] = 0.0f;
            #pragma unroll
            for (int r = 14; r >= 0; --r) en[r] = __builtin_fmaf(hv[r], hv[r], en[r + 1]);
            f32x2 core2 = f32x2{hv[15] * hv[15], 0.0f};
            #pragma unroll
            for (int i = 4; i <= 14; ++i) {
                const f32x4 v = LD4(i);
                core2 = __builtin_elementwise_fma(f32x2{v[0], v[1]}, f32x2{v[0], v[1]}, core2);
                core2 = __builtin_elementwise_fma(f32x2{v[2], v[3]}, f32x2{v[2], v[3]}, core2);
            }
            const f32x4 m = LD4(15);
            core2 = __builtin_elementwise_fma(f32x2{m[0], m[1]}, f32x2{m[0], m[1]}, core2);
            const float core = __builtin_fmaf(m[2], m[2], core2.x + core2.y);
            #pragma unroll
            for (int r = 0; r < 16; ++r) en[r] = en[r] + core;
            const f32x4 t0 = LD4(16), t1 = LD4(17), t2 = LD4(18), t3 = LD4(19);
            const float tv[15] = {m[3], t0[0], t0[1], t0[2], t0[3], t1[0], t1[1], t1[2], t1[3], t2[0], t2[1], t2[2], t2[3], t3[0], t3[1]};
            float tail_run = 0.0f;
            #pragma unroll
            for (int r = 1; r < 16; ++r) { tail_run = __builtin_fmaf(tv[r - 1], tv[r - 1], tail_run); en[r] = en[r] + tail_run; }
        }
        #undef LD4
        float fac[16];
        #pragma unroll
        for (int r = 0; r < 16; ++r) fac[r] = __builtin_amdgcn_rsqf(en[r]);
        if (MODE == 2) matrix();
        wave_fence_lds();
        // the exchange of the experiment: owner layout (lane a, lag r at 20 a + r) -> accumulator layout (tile t, lane (q, c): lags 256 t + 16 c + 4 q + 0..3)
        // through the LDS words of the samples, products written back in place, fetched again in lane order
        #pragma unroll
        for (int r = 0; r < 16; r += 4) *reinterpret_cast<f32x4*>(s + ROW * lane + r) = f32x4{fac[r], fac[r + 1], fac[r + 2], fac[r + 3]};
        wave_fence_lds();
        #pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x4* fp = reinterpret_cast<f32x4*>(s + 16 * ROW * t + ROW * c + 4 * q);
            const f32x4 f = *fp;
            *fp = f32x4{f[0] * 0.5f, f[1] * 0.5f, f[2] * 0.5f, f[3] * 0.5f};
        }
        wave_fence_lds();
        #pragma unroll
        for (int t = 0; t < 4; ++t)
            *reinterpret_cast<f32x4*>(en_out + (long long)item * 1024 + 256 * t + 4 * lane) = *reinterpret_cast<const f32x4*>(s + 16 * ROW * t + 4 * lane + 4 * (lane >> 2));
        #pragma unroll
        for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(acc_out + (long long)item * 1024 + 256 * t + 4 * lane) = acc[t];
        wave_fence_lds();
    }
}

int main()
{
    const long long N = 131072;                     // items of 1 024 lags (+ 64 samples of the next item)
    std::vector<float> h((N + 1) * 1024);
    unsigned x = 12345u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = ((int)(x >> 8) % 20001 - 10000) * 3.0e-5f; }
    float *y, *e, *a;
    CK(hipMalloc(&y, h.size() * 4)); CK(hipMalloc(&e, N * 1024 * 4)); CK(hipMalloc(&a, N * 1024 * 4));
    CK(hipMemcpy(y, h.data(), h.size() * 4, hipMemcpyHostToDevice));
    std::vector<float> ref(N * 1024), cur(N * 1024), accref(N * 1024);
    for (int mode = 0; mode < 3; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipMemset(e, 0, N * 1024 * 4)); CK(hipMemset(a, 0, N * 1024 * 4));
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(768), dim3(256), 0, 0, y, N, e, a);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(768), dim3(256), 0, 0, y, N, e, a);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(768), dim3(256), 0, 0, y, N, e, a);
            CK(hipDeviceSynchronize());
            CK(hipMemcpy(cur.data(), e, N * 1024 * 4, hipMemcpyDeviceToHost));
            if (mode == 0 && rep == 0) ref = cur;
            long long bad = 0, badlane[4] = {0, 0, 0, 0};
            for (long long i = 0; i < N * 1024; ++i) if (memcmp(&cur[i], &ref[i], 4)) { ++bad; ++badlane[(i % 1024) / 256]; }
            long long accbad = -1;
            if (mode) {
                std::vector<float> ac(N * 1024);
                CK(hipMemcpy(ac.data(), a, N * 1024 * 4, hipMemcpyDeviceToHost));
                if (mode == 1 && rep == 0) accref = ac;
                accbad = 0;
                for (long long i = 0; i < N * 1024; ++i) if (memcmp(&ac[i], &accref[i], 4)) ++accbad;
            }
            printf("mode %d launch %d: factor words differing from the matrix-free kernel: %lld (by quarter of the item: %lld %lld %lld %lld); accumulator words differing from the first matrix launch: %lld\n",
                   mode, rep, bad, badlane[0], badlane[1], badlane[2], badlane[3], accbad);
        }
    }
    return 0;
}

// probe: what does v_mfma_f64_16x16x4_f64 compute, bit for bit?  D = A(16x4) x B(4x16) + C per wave.
// (1) operand / result layout, found with one-hot inputs; (2) is D[i][n] the sequential chain fma(a3,b3,fma(a2,b2,fma(a1,b1,fma(a0,b0,c)))),
// the reverse chain, or something else (unfused products, a sum tree)?  Random operands over 40 binades so that the order matters.
//   hipcc -O3 --offload-arch=gfx950 -o ub_mfma_f64 tools/ub/ub_mfma_f64.hip && ./ub_mfma_f64
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>
#include <random>
typedef double double4_t __attribute__((ext_vector_type(4)));

__global__ void k_mfma(const double* A, const double* B, const double* C, double* D, int nprob)
{
    const int lane = threadIdx.x;
    for (int p = blockIdx.x; p < nprob; p += gridDim.x) {
        const double a = A[p * 64 + lane], b = B[p * 64 + lane];
        double4_t c; for (int r = 0; r < 4; ++r) c[r] = C[p * 256 + lane * 4 + r];
        const double4_t d = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
        for (int r = 0; r < 4; ++r) D[p * 256 + lane * 4 + r] = d[r];
    }
}

int main()
{
    // ---- layout: one-hot A and B
    const int NP = 64 * 64;
    std::vector<double> A(NP * 64, 0.0), B(NP * 64, 0.0), C(NP * 256, 0.0), D(NP * 256);
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) { const int p = la * 64 + lb; A[p * 64 + la] = 1.0; B[p * 64 + lb] = 1.0; }
    double *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, NP);
    hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
    // A lane la = (i, k), B lane lb = (k', n): D[i][n] = 1 iff k == k'.  Hypothesis: i = la % 16, k = la / 16; k' = lb / 16, n = lb % 16;
    // result element r of lane l: row l / 16 + 4 r, column l % 16.
    int bad = 0;
    for (int la = 0; la < 64; ++la) for (int lb = 0; lb < 64; ++lb) {
        const int p = la * 64 + lb, i = la % 16, k = la / 16, k2 = lb / 16, n = lb % 16;
        for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
            const int row = (l / 16) + 4 * r, col = l % 16;
            const double want = (k == k2 && row == i && col == n) ? 1.0 : 0.0;
            if (D[p * 256 + l * 4 + r] != want) ++bad;
        }
    }
    printf("layout hypothesis (A lane = i + 16 k, B lane = n + 16 k, D lane l elem r = [l / 16 + 4 r][l %% 16]): %s (%d mismatches)\n", bad ? "WRONG" : "confirmed", bad);
    // ---- arithmetic: random problems
    const int NQ = 4096;
    std::mt19937_64 g(7);
    std::uniform_real_distribution<double> um(-1.0, 1.0); std::uniform_int_distribution<int> ue(-20, 20);
    A.assign(NQ * 64, 0.0); B.assign(NQ * 64, 0.0); C.assign(NQ * 256, 0.0); D.assign(NQ * 256, 0.0);
    for (auto& x : A) x = std::ldexp(um(g), ue(g)); for (auto& x : B) x = std::ldexp(um(g), ue(g)); for (auto& x : C) x = std::ldexp(um(g), ue(g));
    hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dD);
    hipMalloc(&dA, A.size() * 8); hipMalloc(&dB, B.size() * 8); hipMalloc(&dC, C.size() * 8); hipMalloc(&dD, D.size() * 8);
    hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), C.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_mfma, dim3(256), dim3(64), 0, 0, dA, dB, dC, dD, NQ);
    hipMemcpy(D.data(), dD, D.size() * 8, hipMemcpyDeviceToHost);
    long n_fwd = 0, n_rev = 0, n_unf = 0, n_tot = 0, n_tree = 0;
    for (int p = 0; p < NQ; ++p) for (int l = 0; l < 64; ++l) for (int r = 0; r < 4; ++r) {
        const int row = (l / 16) + 4 * r, col = l % 16;
        const double c = C[p * 256 + l * 4 + r];
        double a[4], b[4];
        for (int k = 0; k < 4; ++k) { a[k] = A[p * 64 + row + 16 * k]; b[k] = B[p * 64 + col + 16 * k]; }
        double f = c; for (int k = 0; k < 4; ++k) f = std::fma(a[k], b[k], f);
        double rv = c; for (int k = 3; k >= 0; --k) rv = std::fma(a[k], b[k], rv);
        double un = c; for (int k = 0; k < 4; ++k) un = un + a[k] * b[k];
        const double tr = std::fma(a[0], b[0], a[1] * b[1]) + std::fma(a[2], b[2], a[3] * b[3]) + c;
        const double got = D[p * 256 + l * 4 + r];
        ++n_tot; n_fwd += (got == f); n_rev += (got == rv); n_unf += (got == un); n_tree += (got == tr);
    }
    printf("of %ld results: == ascending fma chain %ld, == descending fma chain %ld, == unfused ascending %ld, == pairwise tree %ld\n", n_tot, n_fwd, n_rev, n_unf, n_tree);
    return 0;
}

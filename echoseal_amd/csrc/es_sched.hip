// es_sched.hip -- key / PN / hop schedule of frame counters, generated on the device (the
// schedule half of f-3).  Replaces, per counter, SecureChannel.pn_bits -> utils.pn_bits -> StreamPRNG.bytes
// (rtwm/crypto.py:46-48, rtwm/utils.py:115-132: ten AES-128-ECB blocks of (ctr << 64 | j), 152 bytes kept) and
// choose_band (rtwm/utils.py:27-36: HMAC-SHA256(key, ctr_be32)[0] % 4).  With it a rank can derive the schedule of
// its own shard from 48 bytes of key material instead of receiving 153 B per counter.
//
// Integer work: one lane per counter (10 AES blocks with the S-box in LDS, two SHA-256 compressions from the
// precomputed inner / outer pad states).  Round keys and pad states are expanded on the host and passed by value.
#include "es_internal.h"
#include <cstring>

namespace {

struct SchedKeys {
    uint32_t rk[44];          // AES-128 round keys, big-endian words (FIPS 197 w[0..43])
    uint32_t ipad[8], opad[8];  // SHA-256 states after the HMAC pad blocks
};

__device__ __forceinline__ uint32_t ror32(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }

__constant__ uint32_t c_K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};

__device__ void sha256_compress(uint32_t st[8], uint32_t w[16])
{
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    #pragma unroll 1
    for (int i = 0; i < 64; ++i) {
        if (i >= 16) {
            const uint32_t w15 = w[(i - 15) & 15], w2 = w[(i - 2) & 15];
            const uint32_t s0 = ror32(w15, 7) ^ ror32(w15, 18) ^ (w15 >> 3);
            const uint32_t s1 = ror32(w2, 17) ^ ror32(w2, 19) ^ (w2 >> 10);
            w[i & 15] = w[i & 15] + s0 + w[(i - 7) & 15] + s1;
        }
        const uint32_t t1 = h + (ror32(e, 6) ^ ror32(e, 11) ^ ror32(e, 25)) + ((e & f) ^ (~e & g)) + c_K256[i] + w[i & 15];
        const uint32_t t2 = (ror32(a, 2) ^ ror32(a, 13) ^ ror32(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

__device__ __forceinline__ uint32_t xtime4(uint32_t x)      // xtime on four packed bytes
{
    return ((x & 0x7f7f7f7fu) << 1) ^ (((x >> 7) & 0x01010101u) * 0x1bu);
}

// columns are big-endian words: byte (row 0) in bits 31..24
__device__ void aes128_encrypt(const SchedKeys& k, const uint8_t* sbox, uint32_t s[4])
{
    #pragma unroll
    for (int c = 0; c < 4; ++c) s[c] ^= k.rk[c];
    #pragma unroll 1
    for (int r = 1; r <= 10; ++r) {
        uint32_t t[4];
        #pragma unroll
        for (int c = 0; c < 4; ++c) {                              // SubBytes + ShiftRows
            t[c] = ((uint32_t)sbox[s[c] >> 24] << 24) | ((uint32_t)sbox[(s[(c + 1) & 3] >> 16) & 255] << 16) |
                   ((uint32_t)sbox[(s[(c + 2) & 3] >> 8) & 255] << 8) | (uint32_t)sbox[s[(c + 3) & 3] & 255];
        }
        if (r < 10) {
            #pragma unroll
            for (int c = 0; c < 4; ++c) {                          // MixColumns on a packed column
                const uint32_t a = t[c], x2 = xtime4(a);
                const uint32_t x3 = x2 ^ a;
                // out_row_i = 2 a_i ^ 3 a_{i+1} ^ a_{i+2} ^ a_{i+3}; rotating left by 8 brings a_{i+1} to row i
                t[c] = x2 ^ ((x3 << 8) | (x3 >> 24)) ^ ((a << 16) | (a >> 16)) ^ ((a << 24) | (a >> 8));
            }
        }
        #pragma unroll
        for (int c = 0; c < 4; ++c) s[c] = t[c] ^ k.rk[4 * r + c];
    }
}

__global__ __launch_bounds__(256) void es_schedule_kernel(SchedKeys k, const uint8_t* __restrict__ sbox_g,
        const uint32_t* __restrict__ ctr_dev, uint32_t ctr0, long long n, uint8_t* __restrict__ pn, uint8_t* __restrict__ band)
{
    __shared__ uint8_t sbox[256];
    sbox[threadIdx.x] = sbox_g[threadIdx.x];
    __syncthreads();
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t ctr = ctr_dev ? ctr_dev[i] : ctr0 + (uint32_t)i;
        uint32_t* row = reinterpret_cast<uint32_t*>(pn + i * ES_PN_BYTES);          // 152-byte rows are 8-byte aligned
        #pragma unroll 1
        for (int j = 0; j < 10; ++j) {
            uint32_t s[4] = {0u, ctr, 0u, (uint32_t)j};                           // (ctr << 64 | j), big endian
            aes128_encrypt(k, sbox, s);
            const int nw = (j < 9) ? 4 : 2;                                          // 152 = 9 * 16 + 8
            for (int c = 0; c < nw; ++c) row[4 * j + c] = __builtin_bswap32(s[c]);
        }
        uint32_t st[8], w[16];
        #pragma unroll
        for (int t = 0; t < 8; ++t) st[t] = k.ipad[t];
        #pragma unroll
        for (int t = 0; t < 16; ++t) w[t] = 0;
        w[0] = ctr; w[1] = 0x80000000u; w[15] = (64 + 4) * 8;                       // message = ctr_be32, after the pad block
        sha256_compress(st, w);
        #pragma unroll
        for (int t = 0; t < 8; ++t) w[t] = st[t];
        w[8] = 0x80000000u;
        #pragma unroll
        for (int t = 9; t < 15; ++t) w[t] = 0;
        w[15] = (64 + 32) * 8;
        #pragma unroll
        for (int t = 0; t < 8; ++t) st[t] = k.opad[t];
        sha256_compress(st, w);
        band[i] = (uint8_t)((st[0] >> 24) & 3u);                                     // tag[0] % 4
    }
}

// ---- host side: S-box, key expansion, HMAC pad states (FIPS 197 / FIPS 180-4 / RFC 2104) ----------------------
uint8_t h_xt(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }
uint8_t h_gmul(uint8_t a, uint8_t b) { uint8_t p = 0; while (b) { if (b & 1) p ^= a; a = h_xt(a); b >>= 1; } return p; }
void host_sbox(uint8_t sb[256])
{
    for (int x = 0; x < 256; ++x) {
        uint8_t inv = 0;
        if (x) for (int yv = 1; yv < 256; ++yv) if (h_gmul((uint8_t)x, (uint8_t)yv) == 1) { inv = (uint8_t)yv; break; }
        uint8_t s = inv, r = inv;
        for (int i = 0; i < 4; ++i) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        sb[x] = (uint8_t)(s ^ 0x63);
    }
}
uint32_t h_ror(uint32_t x, int n) { return (x >> n) | (x << (32 - n)); }
void host_sha256_compress(uint32_t st[8], const uint8_t blk[64])
{
    static const uint32_t K[64] = {
        0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
        0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
        0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
        0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
        0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
        0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
        0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; ++i)
        w[i] = w[i - 16] + (h_ror(w[i - 15], 7) ^ h_ror(w[i - 15], 18) ^ (w[i - 15] >> 3)) + w[i - 7] + (h_ror(w[i - 2], 17) ^ h_ror(w[i - 2], 19) ^ (w[i - 2] >> 10));
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; ++i) {
        const uint32_t t1 = h + (h_ror(e, 6) ^ h_ror(e, 11) ^ h_ror(e, 25)) + ((e & f) ^ (~e & g)) + K[i] + w[i];
        const uint32_t t2 = (h_ror(a, 2) ^ h_ror(a, 13) ^ h_ror(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}

}  // namespace

int es_launch_schedule(es_ctx* ctx, const uint8_t* aes_key16, const uint8_t* band_key32, const uint32_t* ctr_dev,
                       uint32_t ctr0, int64_t n, uint8_t* pn_rows, uint8_t* band, hipStream_t st)
{
    uint8_t sb[256];
    host_sbox(sb);
    if (!ctx->d_sbox) {
        ES_HIP_CHECK(ctx, hipMalloc(&ctx->d_sbox, 256));
        ES_HIP_CHECK(ctx, hipMemcpy(ctx->d_sbox, sb, 256, hipMemcpyHostToDevice));
    }
    SchedKeys k;
    uint8_t rk[176];
    std::memcpy(rk, aes_key16, 16);
    uint8_t rcon = 1;
    for (int i = 16; i < 176; i += 4) {
        uint8_t t[4] = {rk[i - 4], rk[i - 3], rk[i - 2], rk[i - 1]};
        if (i % 16 == 0) {
            const uint8_t t0 = t[0];
            t[0] = (uint8_t)(sb[t[1]] ^ rcon); t[1] = sb[t[2]]; t[2] = sb[t[3]]; t[3] = sb[t0];
            rcon = h_xt(rcon);
        }
        for (int q = 0; q < 4; ++q) rk[i + q] = (uint8_t)(rk[i - 16 + q] ^ t[q]);
    }
    for (int i = 0; i < 44; ++i) k.rk[i] = ((uint32_t)rk[4 * i] << 24) | ((uint32_t)rk[4 * i + 1] << 16) | ((uint32_t)rk[4 * i + 2] << 8) | rk[4 * i + 3];
    static const uint32_t IV[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};
    uint8_t pad[64];
    std::memcpy(k.ipad, IV, sizeof IV); std::memcpy(k.opad, IV, sizeof IV);
    std::memset(pad, 0x36, 64); for (int i = 0; i < 32; ++i) pad[i] ^= band_key32[i];
    host_sha256_compress(k.ipad, pad);
    std::memset(pad, 0x5c, 64); for (int i = 0; i < 32; ++i) pad[i] ^= band_key32[i];
    host_sha256_compress(k.opad, pad);
    long long blocks = (n + 255) / 256;
    const long long cap = (long long)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(es_schedule_kernel, dim3((unsigned)blocks), dim3(256), 0, st, k, ctx->d_sbox, ctr_dev, ctr0, (long long)n,
                       pn_rows, band);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

// es_aead.hip -- batched payload validator and candidate selection: the step after the
// list decoder.  Replaces the Python validator callback the reference hands to PolarCode.decode
// (rtwm/detector.py:168-176 -> SecureChannel.open, rtwm/crypto.py:39-43 -> ChaCha20-Poly1305, RFC 8439) and the
// candidate-selection tail of PolarCode.decode (rtwm/fastpolar.py:268-276, 332-359).
//
// Integer work, a few hundred 32-bit operations per 55-byte blob: one lane per blob for the AEAD check
// (blob = nonce 12 | ciphertext 27 | tag 16, no AAD: Poly1305 runs over three 16-byte blocks), one lane per
// frame for the selection scan.  Blob rows are 55 bytes, so a wave's loads are byte loads from 3 520
// consecutive bytes: every cache line is fetched once.
#include "es_internal.h"

namespace {

struct AeadKey { uint32_t w[8]; };

__device__ __forceinline__ uint32_t rotl32(uint32_t v, int n) { return (v << n) | (v >> (32 - n)); }

#define ES_QR(a, b, c, d)                    \
    a += b; d ^= a; d = rotl32(d, 16);       \
    c += d; b ^= c; b = rotl32(b, 12);       \
    a += b; d ^= a; d = rotl32(d, 8);        \
    c += d; b ^= c; b = rotl32(b, 7);

// RFC 8439 2.3: keystream words of one block
__device__ void chacha20_block(const AeadKey& key, uint32_t counter, const uint32_t nonce[3], uint32_t out[16])
{
    uint32_t s[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
    #pragma unroll
    for (int i = 0; i < 8; ++i) s[4 + i] = key.w[i];
    s[12] = counter; s[13] = nonce[0]; s[14] = nonce[1]; s[15] = nonce[2];
    uint32_t x[16];
    #pragma unroll
    for (int i = 0; i < 16; ++i) x[i] = s[i];
    for (int r = 0; r < 10; ++r) {
        ES_QR(x[0], x[4], x[8], x[12]) ES_QR(x[1], x[5], x[9], x[13]) ES_QR(x[2], x[6], x[10], x[14]) ES_QR(x[3], x[7], x[11], x[15])
        ES_QR(x[0], x[5], x[10], x[15]) ES_QR(x[1], x[6], x[11], x[12]) ES_QR(x[2], x[7], x[8], x[13]) ES_QR(x[3], x[4], x[9], x[14])
    }
    #pragma unroll
    for (int i = 0; i < 16; ++i) out[i] = x[i] + s[i];
}

struct Poly {
    uint32_t r0, r1, r2, r3, r4, s1, s2, s3, s4, h0, h1, h2, h3, h4;
    __device__ void init(const uint32_t k[8])
    {
        // 26-bit limbs of the clamped r (k[0..3] little endian)
        r0 = k[0] & 0x3ffffff;
        r1 = ((k[0] >> 26) | (k[1] << 6)) & 0x3ffff03;
        r2 = ((k[1] >> 20) | (k[2] << 12)) & 0x3ffc0ff;
        r3 = ((k[2] >> 14) | (k[3] << 18)) & 0x3f03fff;
        r4 = (k[3] >> 8) & 0x00fffff;
        s1 = r1 * 5; s2 = r2 * 5; s3 = r3 * 5; s4 = r4 * 5;
        h0 = h1 = h2 = h3 = h4 = 0;
    }
    // one full 16-byte block m[0..3] (little-endian words), with the 2^128 bit
    __device__ void block(const uint32_t m[4])
    {
        h0 += m[0] & 0x3ffffff;
        h1 += ((m[0] >> 26) | (m[1] << 6)) & 0x3ffffff;
        h2 += ((m[1] >> 20) | (m[2] << 12)) & 0x3ffffff;
        h3 += ((m[2] >> 14) | (m[3] << 18)) & 0x3ffffff;
        h4 += (m[3] >> 8) | (1u << 24);
        const uint64_t d0 = (uint64_t)h0 * r0 + (uint64_t)h1 * s4 + (uint64_t)h2 * s3 + (uint64_t)h3 * s2 + (uint64_t)h4 * s1;
        uint64_t d1 = (uint64_t)h0 * r1 + (uint64_t)h1 * r0 + (uint64_t)h2 * s4 + (uint64_t)h3 * s3 + (uint64_t)h4 * s2;
        uint64_t d2 = (uint64_t)h0 * r2 + (uint64_t)h1 * r1 + (uint64_t)h2 * r0 + (uint64_t)h3 * s4 + (uint64_t)h4 * s3;
        uint64_t d3 = (uint64_t)h0 * r3 + (uint64_t)h1 * r2 + (uint64_t)h2 * r1 + (uint64_t)h3 * r0 + (uint64_t)h4 * s4;
        uint64_t d4 = (uint64_t)h0 * r4 + (uint64_t)h1 * r3 + (uint64_t)h2 * r2 + (uint64_t)h3 * r1 + (uint64_t)h4 * r0;
        uint32_t c = (uint32_t)(d0 >> 26); h0 = (uint32_t)d0 & 0x3ffffff;
        d1 += c; c = (uint32_t)(d1 >> 26); h1 = (uint32_t)d1 & 0x3ffffff;
        d2 += c; c = (uint32_t)(d2 >> 26); h2 = (uint32_t)d2 & 0x3ffffff;
        d3 += c; c = (uint32_t)(d3 >> 26); h3 = (uint32_t)d3 & 0x3ffffff;
        d4 += c; c = (uint32_t)(d4 >> 26); h4 = (uint32_t)d4 & 0x3ffffff;
        h0 += c * 5; c = h0 >> 26; h0 &= 0x3ffffff; h1 += c;
    }
    // tag = (h mod p + s) mod 2^128, s = k[4..7]
    __device__ void finish(const uint32_t k[8], uint32_t tag[4])
    {
        uint32_t c = h1 >> 26; h1 &= 0x3ffffff;
        h2 += c; c = h2 >> 26; h2 &= 0x3ffffff;
        h3 += c; c = h3 >> 26; h3 &= 0x3ffffff;
        h4 += c; c = h4 >> 26; h4 &= 0x3ffffff;
        h0 += c * 5; c = h0 >> 26; h0 &= 0x3ffffff; h1 += c;
        uint32_t g0 = h0 + 5; c = g0 >> 26; g0 &= 0x3ffffff;
        uint32_t g1 = h1 + c; c = g1 >> 26; g1 &= 0x3ffffff;
        uint32_t g2 = h2 + c; c = g2 >> 26; g2 &= 0x3ffffff;
        uint32_t g3 = h3 + c; c = g3 >> 26; g3 &= 0x3ffffff;
        const uint32_t g4 = h4 + c - (1u << 26);
        const uint32_t mask = (g4 >> 31) - 1;                      // all ones if h >= p
        h0 = (h0 & ~mask) | (g0 & mask); h1 = (h1 & ~mask) | (g1 & mask); h2 = (h2 & ~mask) | (g2 & mask);
        h3 = (h3 & ~mask) | (g3 & mask); h4 = (h4 & ~mask) | (g4 & mask);
        const uint32_t w0 = h0 | (h1 << 26), w1 = (h1 >> 6) | (h2 << 20), w2 = (h2 >> 12) | (h3 << 14), w3 = (h3 >> 18) | (h4 << 8);
        uint64_t f = (uint64_t)w0 + k[4]; tag[0] = (uint32_t)f;
        f = (uint64_t)w1 + k[5] + (f >> 32); tag[1] = (uint32_t)f;
        f = (uint64_t)w2 + k[6] + (f >> 32); tag[2] = (uint32_t)f;
        f = (uint64_t)w3 + k[7] + (f >> 32); tag[3] = (uint32_t)f;
    }
};

// The detector's validator on one blob: tag verifies, plaintext starts with "ESAL", bytes 4..7 (big endian) == ctr.
// `plain` (nullable) receives the 27 plaintext bytes when the tag verifies, zeros otherwise.
__device__ bool validate_blob(const AeadKey& key, const uint8_t* __restrict__ blob, uint32_t ctr, uint8_t* plain)
{
    uint32_t w[14];                                                // the 55 bytes as little-endian words (+1 pad byte)
    #pragma unroll
    for (int i = 0; i < 14; ++i) {
        uint32_t v = 0;
        #pragma unroll
        for (int b = 0; b < 4; ++b) { const int o = 4 * i + b; if (o < ES_INFO_BYTES) v |= (uint32_t)blob[o] << (8 * b); }
        w[i] = v;
    }
    const uint32_t nonce[3] = {w[0], w[1], w[2]};
    // ciphertext = bytes 12..38 -> words w[3..9] (27 bytes: w[9] keeps its low 3 bytes); tag = bytes 39..54
    uint32_t ct[8];
    #pragma unroll
    for (int i = 0; i < 6; ++i) ct[i] = w[3 + i];
    ct[6] = w[9] & 0x00ffffffu; ct[7] = 0;
    uint32_t tag[4];
    #pragma unroll
    for (int i = 0; i < 4; ++i) tag[i] = (w[9 + i] >> 24) | (w[10 + i] << 8);
    uint32_t ks[16];
    chacha20_block(key, 0, nonce, ks);                             // one-time Poly1305 key = first 32 bytes
    Poly P; P.init(ks);
    P.block(ct); P.block(ct + 4);                                  // ciphertext padded to 32 bytes
    const uint32_t lens[4] = {0u, 0u, 27u, 0u};                    // le64(len(aad) = 0) | le64(len(ct) = 27)
    P.block(lens);
    uint32_t mac[4];
    P.finish(ks, mac);
    const bool tag_ok = ((mac[0] ^ tag[0]) | (mac[1] ^ tag[1]) | (mac[2] ^ tag[2]) | (mac[3] ^ tag[3])) == 0;
    uint32_t pt[7];
    chacha20_block(key, 1, nonce, ks);
    #pragma unroll
    for (int i = 0; i < 7; ++i) pt[i] = tag_ok ? (ct[i] ^ ks[i]) : 0u;
    pt[6] &= 0x00ffffffu;
    if (plain) {
        #pragma unroll
        for (int i = 0; i < 27; ++i) plain[i] = (uint8_t)(pt[i >> 2] >> (8 * (i & 3)));
    }
    const uint32_t be_ctr = __builtin_bswap32(pt[1]);
    return tag_ok && pt[0] == 0x4c415345u /* "ESAL" little endian */ && be_ctr == ctr;
}

// SecureChannel.seal (rtwm/crypto.py:33-37): blob = nonce 12 | ChaCha20(counter 1) xor plaintext 27 | Poly1305 tag 16
__global__ __launch_bounds__(256) void es_aead_seal_kernel(AeadKey key, const uint8_t* __restrict__ nonces,
        const uint8_t* __restrict__ plain, long long n, uint8_t* __restrict__ blobs)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint8_t* nb = nonces + i * 12;
        const uint8_t* pb = plain + i * 27;
        uint32_t nonce[3], pt[8];
        #pragma unroll
        for (int w = 0; w < 3; ++w) nonce[w] = (uint32_t)nb[4 * w] | ((uint32_t)nb[4 * w + 1] << 8) | ((uint32_t)nb[4 * w + 2] << 16) | ((uint32_t)nb[4 * w + 3] << 24);
        #pragma unroll
        for (int w = 0; w < 8; ++w) {
            uint32_t v = 0;
            #pragma unroll
            for (int b = 0; b < 4; ++b) { const int o = 4 * w + b; if (o < 27) v |= (uint32_t)pb[o] << (8 * b); }
            pt[w] = v;
        }
        uint32_t ks[16], ct[8];
        chacha20_block(key, 1, nonce, ks);
        #pragma unroll
        for (int w = 0; w < 7; ++w) ct[w] = pt[w] ^ ks[w];
        ct[6] &= 0x00ffffffu; ct[7] = 0;
        chacha20_block(key, 0, nonce, ks);
        Poly P; P.init(ks);
        P.block(ct); P.block(ct + 4);
        const uint32_t lens[4] = {0u, 0u, 27u, 0u};
        P.block(lens);
        uint32_t tag[4];
        P.finish(ks, tag);
        uint8_t* out = blobs + i * ES_INFO_BYTES;
        for (int b = 0; b < 12; ++b) out[b] = nb[b];
        for (int b = 0; b < 27; ++b) out[12 + b] = (uint8_t)(ct[b >> 2] >> (8 * (b & 3)));
        for (int b = 0; b < 16; ++b) out[39 + b] = (uint8_t)(tag[b >> 2] >> (8 * (b & 3)));
    }
}

__global__ __launch_bounds__(256) void es_aead_check_kernel(AeadKey key, const uint8_t* __restrict__ blobs, long long n,
        int group, const uint32_t* __restrict__ ctr, uint8_t* __restrict__ ok, uint8_t* __restrict__ plain)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride)
        ok[i] = validate_blob(key, blobs + i * ES_INFO_BYTES, ctr[i / group], plain ? plain + i * 27 : nullptr) ? 1 : 0;
}

// PolarCode.decode's selection (rtwm/fastpolar.py:268-276, 332-359) for one frame per lane.
__global__ __launch_bounds__(256) void es_select_kernel(AeadKey key, int use_key, const uint32_t* __restrict__ ctr, long long B, int L,
        const uint8_t* __restrict__ hard_info, const uint8_t* __restrict__ hard_ok, const uint8_t* __restrict__ cand_info,
        const double* __restrict__ cand_metric, const uint8_t* __restrict__ cand_ok, const int32_t* __restrict__ ncand,
        uint8_t* __restrict__ payload, int8_t* __restrict__ ok_out, int32_t* __restrict__ which_out)
{
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long f = (long long)blockIdx.x * blockDim.x + threadIdx.x; f < B; f += stride) {
        const uint32_t c = use_key ? ctr[f] : 0u;
        const uint8_t* src = hard_info + f * ES_INFO_BYTES;
        int which = -1, ok = 0;
        const int n = ncand[f];
        if (n < 0) ok = -2;                                        // the list decoder could not decode this record (es_scl_batch): nothing of its rows is defined
        else if (hard_ok[f] && (!use_key || validate_blob(key, src, c, nullptr))) ok = 1;
        else {
            if (n == 0) ok = -1;                                   // list loop was skipped: usage error, reported to the host
            else {
                int best_crc = -1, best_any = -1;
                double best_any_m = __builtin_inf();
                const uint8_t* ci = cand_info + f * (long long)L * ES_INFO_BYTES;
                const double* cm = cand_metric + f * (long long)L;
                const uint8_t* co = cand_ok + f * (long long)L;
                for (int r = 0; r < n && !ok; ++r) {
                    if (co[r]) {
                        if (!use_key || validate_blob(key, ci + (long long)r * ES_INFO_BYTES, c, nullptr)) { which = r; ok = 1; }
                        else if (best_crc < 0 || cm[r] < cm[best_crc]) best_crc = r;
                    } else if (cm[r] < best_any_m) { best_any = r; best_any_m = cm[r]; }
                }
                if (!ok) which = best_crc >= 0 ? best_crc : best_any;
                if (which >= 0) src = ci + (long long)which * ES_INFO_BYTES;
            }
        }
        for (int k = 0; k < ES_INFO_BYTES; ++k) payload[f * ES_INFO_BYTES + k] = (ok == -2) ? (uint8_t)0 : src[k];
        ok_out[f] = (int8_t)ok;
        which_out[f] = which;
    }
}

AeadKey load_key(const uint8_t* key32)
{
    AeadKey k;
    for (int i = 0; i < 8; ++i)
        k.w[i] = (uint32_t)key32[4 * i] | ((uint32_t)key32[4 * i + 1] << 8) | ((uint32_t)key32[4 * i + 2] << 16) | ((uint32_t)key32[4 * i + 3] << 24);
    return k;
}

}  // namespace

int es_launch_aead_check(es_ctx* ctx, const uint8_t* key32, const uint8_t* blobs, int64_t n, int group, const uint32_t* ctr,
                         uint8_t* ok, uint8_t* plain, hipStream_t st)
{
    long long blocks = (n + 255) / 256;
    const long long cap = (long long)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(es_aead_check_kernel, dim3((unsigned)blocks), dim3(256), 0, st, load_key(key32), blobs, (long long)n,
                       group, ctr, ok, plain);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

int es_launch_aead_seal(es_ctx* ctx, const uint8_t* key32, const uint8_t* nonces, const uint8_t* plain, int64_t n, uint8_t* blobs,
                        hipStream_t st)
{
    long long blocks = (n + 255) / 256;
    const long long cap = (long long)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL(es_aead_seal_kernel, dim3((unsigned)blocks), dim3(256), 0, st, load_key(key32), nonces, plain, (long long)n, blobs);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

int es_launch_select(es_ctx* ctx, const uint8_t* key32, const uint32_t* ctr, int64_t B, int L, const uint8_t* hard_info,
                     const uint8_t* hard_ok, const uint8_t* cand_info, const double* cand_metric, const uint8_t* cand_ok,
                     const int32_t* ncand, uint8_t* payload, int8_t* ok, int32_t* which, hipStream_t st)
{
    long long blocks = (B + 255) / 256;
    const long long cap = (long long)ctx->num_cu * 8;
    if (blocks > cap) blocks = cap;
    AeadKey k{};
    if (key32) k = load_key(key32);
    hipLaunchKernelGGL(es_select_kernel, dim3((unsigned)blocks), dim3(256), 0, st, k, key32 ? 1 : 0, ctr, (long long)B, L,
                       hard_info, hard_ok, cand_info, cand_metric, cand_ok, ncand, payload, ok, which);
    ES_HIP_CHECK(ctx, hipGetLastError());
    return ES_OK;
}

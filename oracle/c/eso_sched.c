/* eso_sched.c -- CPU ORACLE (test infrastructure only) for the key/PN/hop schedule of a frame counter
 * (on-device generation = the schedule half of f-3).
 *
 * What it restates:
 *   StreamPRNG.bytes / pn_bits   rtwm/utils.py:115-132   block j of frame c = AES-128-ECB(sub_key, (c << 64 | j) as 16
 *                                                        big-endian bytes); the PN bits are those bytes, MSB first
 *   SecureChannel.pn_bits         rtwm/crypto.py:46-48    (same stream; the detector consumes 1 215 bits = 152 bytes)
 *   choose_band                   rtwm/utils.py:27-36     HMAC-SHA256(key, ctr as 4 big-endian bytes)[0] % 4
 * AES-128 and SHA-256 live in third-party packages absent from /root/reference (PyCryptodome / cryptography /
 * hashlib); the published algorithms are restated: FIPS 197, FIPS 180-4, RFC 2104.  Pinned by FIPS 197 C.1,
 * RFC 4231 test case 2 and the reference's own known answers for key 0xAA*32 (band indices of
 * counters 0..15, pn_bits(0,128), pn_bits(5,32), sha256 of pn_bits(5,1215)) in tests/test_oracle_sched.py.
 */
#include <stdint.h>
#include <string.h>

/* ---------------------------------------------------------------- AES-128 (FIPS 197) */
static uint8_t SBOX[256];
static int sbox_ready = 0;
static uint8_t xt(uint8_t x) { return (uint8_t)((x << 1) ^ ((x >> 7) * 0x1b)); }
static uint8_t gmul(uint8_t a, uint8_t b) { uint8_t p = 0; while (b) { if (b & 1) p ^= a; a = xt(a); b >>= 1; } return p; }
static void build_sbox(void)
{
    if (sbox_ready) return;
    for (int x = 0; x < 256; ++x) {
        uint8_t inv = 0;
        if (x) for (int yv = 1; yv < 256; ++yv) if (gmul((uint8_t)x, (uint8_t)yv) == 1) { inv = (uint8_t)yv; break; }
        uint8_t s = inv, r = inv;
        for (int i = 0; i < 4; ++i) { r = (uint8_t)((r << 1) | (r >> 7)); s ^= r; }
        SBOX[x] = (uint8_t)(s ^ 0x63);
    }
    sbox_ready = 1;
}
void eso_aes128_expand(const uint8_t key[16], uint8_t rk[176])
{
    build_sbox();
    memcpy(rk, key, 16);
    uint8_t rcon = 1;
    for (int i = 16; i < 176; i += 4) {
        uint8_t t[4] = {rk[i - 4], rk[i - 3], rk[i - 2], rk[i - 1]};
        if (i % 16 == 0) {
            const uint8_t t0 = t[0];
            t[0] = (uint8_t)(SBOX[t[1]] ^ rcon); t[1] = SBOX[t[2]]; t[2] = SBOX[t[3]]; t[3] = SBOX[t0];
            rcon = xt(rcon);
        }
        for (int k = 0; k < 4; ++k) rk[i + k] = (uint8_t)(rk[i - 16 + k] ^ t[k]);
    }
}
void eso_aes128_encrypt(const uint8_t rk[176], const uint8_t in[16], uint8_t out[16])
{
    uint8_t s[16];
    for (int i = 0; i < 16; ++i) s[i] = (uint8_t)(in[i] ^ rk[i]);
    for (int r = 1; r <= 10; ++r) {
        uint8_t t[16];
        for (int c = 0; c < 4; ++c) for (int row = 0; row < 4; ++row) t[4 * c + row] = SBOX[s[4 * ((c + row) & 3) + row]];   /* SubBytes + ShiftRows */
        if (r < 10) {
            for (int c = 0; c < 4; ++c) {
                const uint8_t a0 = t[4 * c], a1 = t[4 * c + 1], a2 = t[4 * c + 2], a3 = t[4 * c + 3];
                s[4 * c]     = (uint8_t)(xt(a0) ^ (xt(a1) ^ a1) ^ a2 ^ a3);
                s[4 * c + 1] = (uint8_t)(a0 ^ xt(a1) ^ (xt(a2) ^ a2) ^ a3);
                s[4 * c + 2] = (uint8_t)(a0 ^ a1 ^ xt(a2) ^ (xt(a3) ^ a3));
                s[4 * c + 3] = (uint8_t)((xt(a0) ^ a0) ^ a1 ^ a2 ^ xt(a3));
            }
        } else memcpy(s, t, 16);
        for (int i = 0; i < 16; ++i) s[i] ^= rk[16 * r + i];
    }
    memcpy(out, s, 16);
}

/* ---------------------------------------------------------------- SHA-256 (FIPS 180-4), HMAC (RFC 2104) */
static const uint32_t K256[64] = {
    0x428a2f98, 0x71374491, 0xb5c0fbcf, 0xe9b5dba5, 0x3956c25b, 0x59f111f1, 0x923f82a4, 0xab1c5ed5, 0xd807aa98, 0x12835b01,
    0x243185be, 0x550c7dc3, 0x72be5d74, 0x80deb1fe, 0x9bdc06a7, 0xc19bf174, 0xe49b69c1, 0xefbe4786, 0x0fc19dc6, 0x240ca1cc,
    0x2de92c6f, 0x4a7484aa, 0x5cb0a9dc, 0x76f988da, 0x983e5152, 0xa831c66d, 0xb00327c8, 0xbf597fc7, 0xc6e00bf3, 0xd5a79147,
    0x06ca6351, 0x14292967, 0x27b70a85, 0x2e1b2138, 0x4d2c6dfc, 0x53380d13, 0x650a7354, 0x766a0abb, 0x81c2c92e, 0x92722c85,
    0xa2bfe8a1, 0xa81a664b, 0xc24b8b70, 0xc76c51a3, 0xd192e819, 0xd6990624, 0xf40e3585, 0x106aa070, 0x19a4c116, 0x1e376c08,
    0x2748774c, 0x34b0bcb5, 0x391c0cb3, 0x4ed8aa4a, 0x5b9cca4f, 0x682e6ff3, 0x748f82ee, 0x78a5636f, 0x84c87814, 0x8cc70208,
    0x90befffa, 0xa4506ceb, 0xbef9a3f7, 0xc67178f2};
#define ROR(x, n) (((x) >> (n)) | ((x) << (32 - (n))))
void eso_sha256_compress(uint32_t st[8], const uint8_t blk[64])
{
    uint32_t w[64];
    for (int i = 0; i < 16; ++i) w[i] = ((uint32_t)blk[4 * i] << 24) | ((uint32_t)blk[4 * i + 1] << 16) | ((uint32_t)blk[4 * i + 2] << 8) | blk[4 * i + 3];
    for (int i = 16; i < 64; ++i) {
        const uint32_t s0 = ROR(w[i - 15], 7) ^ ROR(w[i - 15], 18) ^ (w[i - 15] >> 3);
        const uint32_t s1 = ROR(w[i - 2], 17) ^ ROR(w[i - 2], 19) ^ (w[i - 2] >> 10);
        w[i] = w[i - 16] + s0 + w[i - 7] + s1;
    }
    uint32_t a = st[0], b = st[1], c = st[2], d = st[3], e = st[4], f = st[5], g = st[6], h = st[7];
    for (int i = 0; i < 64; ++i) {
        const uint32_t t1 = h + (ROR(e, 6) ^ ROR(e, 11) ^ ROR(e, 25)) + ((e & f) ^ (~e & g)) + K256[i] + w[i];
        const uint32_t t2 = (ROR(a, 2) ^ ROR(a, 13) ^ ROR(a, 22)) + ((a & b) ^ (a & c) ^ (b & c));
        h = g; g = f; f = e; e = d + t1; d = c; c = b; b = a; a = t1 + t2;
    }
    st[0] += a; st[1] += b; st[2] += c; st[3] += d; st[4] += e; st[5] += f; st[6] += g; st[7] += h;
}
static const uint32_t IV256[8] = {0x6a09e667, 0xbb67ae85, 0x3c6ef372, 0xa54ff53a, 0x510e527f, 0x9b05688c, 0x1f83d9ab, 0x5be0cd19};

/* HMAC-SHA256 for keys <= 64 bytes and messages <= 55 bytes (one inner block after the pad block) */
void eso_hmac_sha256_short(const uint8_t* key, int klen, const uint8_t* msg, int mlen, uint8_t out[32])
{
    uint8_t pad[64], blk[64];
    uint32_t in[8], ou[8];
    memcpy(in, IV256, sizeof in); memcpy(ou, IV256, sizeof ou);
    memset(pad, 0x36, 64); for (int i = 0; i < klen; ++i) pad[i] ^= key[i];
    eso_sha256_compress(in, pad);
    memset(pad, 0x5c, 64); for (int i = 0; i < klen; ++i) pad[i] ^= key[i];
    eso_sha256_compress(ou, pad);
    memset(blk, 0, 64); memcpy(blk, msg, (size_t)mlen); blk[mlen] = 0x80;
    const uint64_t bits = (uint64_t)(64 + mlen) * 8;
    for (int i = 0; i < 8; ++i) blk[63 - i] = (uint8_t)(bits >> (8 * i));
    eso_sha256_compress(in, blk);
    memset(blk, 0, 64);
    for (int i = 0; i < 8; ++i) { blk[4 * i] = (uint8_t)(in[i] >> 24); blk[4 * i + 1] = (uint8_t)(in[i] >> 16); blk[4 * i + 2] = (uint8_t)(in[i] >> 8); blk[4 * i + 3] = (uint8_t)in[i]; }
    blk[32] = 0x80; blk[62] = 0x03; blk[63] = 0x00;             /* (64 + 32) * 8 = 768 = 0x0300 bits */
    eso_sha256_compress(ou, blk);
    for (int i = 0; i < 8; ++i) { out[4 * i] = (uint8_t)(ou[i] >> 24); out[4 * i + 1] = (uint8_t)(ou[i] >> 16); out[4 * i + 2] = (uint8_t)(ou[i] >> 8); out[4 * i + 3] = (uint8_t)ou[i]; }
}

/* Schedule row of frame counter ctr: 152 packed PN bytes (rtwm/utils.py:115-132) and the band index
 * (rtwm/utils.py:27-36).  aes_key = the 16-byte PN sub-key, band_key = the 32-byte hop key. */
void eso_schedule_row(const uint8_t aes_key[16], const uint8_t band_key[32], uint32_t ctr, uint8_t pn152[152], uint8_t* band)
{
    uint8_t rk[176], in[16], out[16], tag[32], msg[4];
    eso_aes128_expand(aes_key, rk);
    for (int j = 0; j < 10; ++j) {
        memset(in, 0, 16);
        in[4] = (uint8_t)(ctr >> 24); in[5] = (uint8_t)(ctr >> 16); in[6] = (uint8_t)(ctr >> 8); in[7] = (uint8_t)ctr;   /* c << 64, c < 2^32 */
        in[15] = (uint8_t)j;
        eso_aes128_encrypt(rk, in, out);
        const int nb = (j < 9) ? 16 : 152 - 144;
        memcpy(pn152 + 16 * j, out, (size_t)nb);
    }
    msg[0] = (uint8_t)(ctr >> 24); msg[1] = (uint8_t)(ctr >> 16); msg[2] = (uint8_t)(ctr >> 8); msg[3] = (uint8_t)ctr;
    eso_hmac_sha256_short(band_key, 32, msg, 4, tag);
    *band = (uint8_t)(tag[0] % 4);
}

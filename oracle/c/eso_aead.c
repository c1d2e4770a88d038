/* eso_aead.c -- CPU ORACLE (test infrastructure, never shipped or measured as the product) for the
 * payload validator of the reference's decode path.
 *
 * What it restates:
 *   SecureChannel.open             rtwm/crypto.py:39-43   blob = nonce(12) || ciphertext || tag(16);
 *                                                         ChaCha20Poly1305(aead_key).decrypt(nonce, ct||tag, b"")
 *   the validator closure          rtwm/detector.py:168-176  open ok, plaintext starts with b"ESAL",
 *                                                         plaintext[4:8] big-endian == frame counter
 *   candidate selection            rtwm/fastpolar.py:268-276,332-359  (hard candidate, then the list in metric
 *                                                         order; exceptions inside the validator count as False)
 *
 * The AEAD itself lives in a third-party dependency that is absent from /root/reference (the `cryptography`
 * package, reference pins only "cryptography>=42" in pyproject; not installed in this image), so the
 * published algorithm is restated here: RFC 8439 (ChaCha20 section 2.3/2.4, Poly1305 section 2.5, AEAD
 * construction section 2.8).  Pinned by the RFC's own vectors (tests/test_oracle_aead.py: 2.3.2 block,
 * 2.5.2 tag, 2.8.2 AEAD) and by blobs sealed through the reference's own call site with the host
 * primitives (which carry the same RFC vectors, tests/test_primitives.py).
 */
#include <stdint.h>
#include <string.h>

#define ROTL32(v, n) (((v) << (n)) | ((v) >> (32 - (n))))
#define QR(a, b, c, d)                       \
    a += b; d ^= a; d = ROTL32(d, 16);       \
    c += d; b ^= c; b = ROTL32(b, 12);       \
    a += b; d ^= a; d = ROTL32(d, 8);        \
    c += d; b ^= c; b = ROTL32(b, 7);

static uint32_t le32(const uint8_t* p) { return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24); }

/* RFC 8439 2.3: one 64-byte keystream block */
void eso_chacha20_block(const uint8_t key[32], uint32_t counter, const uint8_t nonce[12], uint8_t out[64])
{
    uint32_t s[16], x[16];
    s[0] = 0x61707865u; s[1] = 0x3320646eu; s[2] = 0x79622d32u; s[3] = 0x6b206574u;
    for (int i = 0; i < 8; ++i) s[4 + i] = le32(key + 4 * i);
    s[12] = counter;
    for (int i = 0; i < 3; ++i) s[13 + i] = le32(nonce + 4 * i);
    memcpy(x, s, sizeof x);
    for (int r = 0; r < 10; ++r) {
        QR(x[0], x[4], x[8], x[12]) QR(x[1], x[5], x[9], x[13]) QR(x[2], x[6], x[10], x[14]) QR(x[3], x[7], x[11], x[15])
        QR(x[0], x[5], x[10], x[15]) QR(x[1], x[6], x[11], x[12]) QR(x[2], x[7], x[8], x[13]) QR(x[3], x[4], x[9], x[14])
    }
    for (int i = 0; i < 16; ++i) {
        const uint32_t v = x[i] + s[i];
        out[4 * i] = (uint8_t)v; out[4 * i + 1] = (uint8_t)(v >> 8); out[4 * i + 2] = (uint8_t)(v >> 16); out[4 * i + 3] = (uint8_t)(v >> 24);
    }
}

/* RFC 8439 2.5: Poly1305 with 26-bit limbs.  msg is processed in 16-byte blocks, the last one may be short. */
void eso_poly1305(const uint8_t otk[32], const uint8_t* msg, size_t len, uint8_t tag[16])
{
    const uint32_t r0 = le32(otk) & 0x3ffffff, r1 = (le32(otk + 3) >> 2) & 0x3ffff03, r2 = (le32(otk + 6) >> 4) & 0x3ffc0ff,
                   r3 = (le32(otk + 9) >> 6) & 0x3f03fff, r4 = (le32(otk + 12) >> 8) & 0x00fffff;
    const uint32_t s1 = r1 * 5, s2 = r2 * 5, s3 = r3 * 5, s4 = r4 * 5;
    uint32_t h0 = 0, h1 = 0, h2 = 0, h3 = 0, h4 = 0;
    while (len > 0) {
        uint8_t blk[17];
        const size_t n = len < 16 ? len : 16;
        memset(blk, 0, sizeof blk);
        memcpy(blk, msg, n);
        blk[n] = 1;                                   /* the 2^(8n) bit */
        h0 += le32(blk) & 0x3ffffff;
        h1 += (le32(blk + 3) >> 2) & 0x3ffffff;
        h2 += (le32(blk + 6) >> 4) & 0x3ffffff;
        h3 += (le32(blk + 9) >> 6) & 0x3ffffff;
        h4 += (le32(blk + 12) >> 8) | ((uint32_t)blk[16] << 24);
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
        msg += n; len -= n;
    }
    /* full carry, then h - p, select */
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
    const uint32_t mask = (g4 >> 31) - 1;             /* all ones if h >= p */
    h0 = (h0 & ~mask) | (g0 & mask); h1 = (h1 & ~mask) | (g1 & mask); h2 = (h2 & ~mask) | (g2 & mask);
    h3 = (h3 & ~mask) | (g3 & mask); h4 = (h4 & ~mask) | (g4 & mask);
    const uint32_t w0 = h0 | (h1 << 26), w1 = (h1 >> 6) | (h2 << 20), w2 = (h2 >> 12) | (h3 << 14), w3 = (h3 >> 18) | (h4 << 8);
    uint64_t f = (uint64_t)w0 + le32(otk + 16);
    uint32_t o[4];
    o[0] = (uint32_t)f; f = (uint64_t)w1 + le32(otk + 20) + (f >> 32);
    o[1] = (uint32_t)f; f = (uint64_t)w2 + le32(otk + 24) + (f >> 32);
    o[2] = (uint32_t)f; f = (uint64_t)w3 + le32(otk + 28) + (f >> 32);
    o[3] = (uint32_t)f;
    for (int i = 0; i < 4; ++i) { tag[4 * i] = (uint8_t)o[i]; tag[4 * i + 1] = (uint8_t)(o[i] >> 8); tag[4 * i + 2] = (uint8_t)(o[i] >> 16); tag[4 * i + 3] = (uint8_t)(o[i] >> 24); }
}

/* RFC 8439 2.8: returns 1 and writes ctlen plaintext bytes if the tag verifies, else 0 (out untouched). */
int eso_aead_open(const uint8_t key[32], const uint8_t nonce[12], const uint8_t* aad, size_t aadlen,
                  const uint8_t* ct, size_t ctlen, const uint8_t tag[16], uint8_t* out)
{
    uint8_t ks[64], mac[16];
    uint8_t buf[4096];
    const size_t pa = (16 - aadlen % 16) % 16, pc = (16 - ctlen % 16) % 16;
    const size_t total = aadlen + pa + ctlen + pc + 16;
    if (total > sizeof buf) return -1;
    eso_chacha20_block(key, 0, nonce, ks);
    memset(buf, 0, total);
    memcpy(buf, aad, aadlen);
    memcpy(buf + aadlen + pa, ct, ctlen);
    uint8_t* lens = buf + aadlen + pa + ctlen + pc;
    for (int i = 0; i < 8; ++i) { lens[i] = (uint8_t)((uint64_t)aadlen >> (8 * i)); lens[8 + i] = (uint8_t)((uint64_t)ctlen >> (8 * i)); }
    eso_poly1305(ks, buf, total, mac);
    uint8_t diff = 0;
    for (int i = 0; i < 16; ++i) diff |= mac[i] ^ tag[i];
    if (diff) return 0;
    for (size_t off = 0, blk = 1; off < ctlen; off += 64, ++blk) {
        eso_chacha20_block(key, (uint32_t)blk, nonce, ks);
        for (size_t i = 0; i < 64 && off + i < ctlen; ++i) out[off + i] = ct[off + i] ^ ks[i];
    }
    return 1;
}

/* The detector's validator on one 55-byte blob (rtwm/detector.py:168-176 through rtwm/crypto.py:39-43).
 * Returns 1 iff the tag verifies, the 27-byte plaintext starts with "ESAL" and bytes 4..7 (big endian)
 * equal `ctr`.  plain27 (nullable) receives the plaintext when the tag verifies, zeros otherwise. */
int eso_validate_blob(const uint8_t key[32], const uint8_t blob[55], uint32_t ctr, uint8_t* plain27)
{
    uint8_t pt[27];
    memset(pt, 0, sizeof pt);
    const int ok = eso_aead_open(key, blob, (const uint8_t*)"", 0, blob + 12, 27, blob + 39, pt);
    if (plain27) memcpy(plain27, pt, 27);
    if (ok != 1) return 0;
    if (memcmp(pt, "ESAL", 4) != 0) return 0;
    const uint32_t c = ((uint32_t)pt[4] << 24) | ((uint32_t)pt[5] << 16) | ((uint32_t)pt[6] << 8) | pt[7];
    return c == ctr;
}

/* Candidate selection of PolarCode.decode (rtwm/fastpolar.py:268-276,332-359), one frame: hard candidate first,
 * then the list in the order given (ascending metric).  key == NULL means validator=None.
 * which = -1 (hard candidate), k (list index).  Returns the reference's ok flag, or -1 when the list is empty
 * although the shortcut did not return (the caller skipped the list loop: a usage error). */
int eso_select_validated(const uint8_t* key, uint32_t ctr, const uint8_t hard[55], int hard_crc_ok,
                         const uint8_t* cand /*[n][55]*/, const uint8_t* cand_crc_ok, const double* cand_metric,
                         int n, uint8_t payload[55], int* which)
{
    memcpy(payload, hard, 55); *which = -1;
    if (hard_crc_ok && (!key || eso_validate_blob(key, hard, ctr, 0))) return 1;
    if (n <= 0) return -1;
    int best_crc = -1, best_any = -1;
    double best_any_m = __builtin_inf();
    for (int r = 0; r < n; ++r) {
        if (cand_crc_ok[r]) {
            if (!key || eso_validate_blob(key, cand + 55 * (size_t)r, ctr, 0)) { memcpy(payload, cand + 55 * (size_t)r, 55); *which = r; return 1; }
            if (best_crc < 0 || cand_metric[r] < cand_metric[best_crc]) best_crc = r;
        } else if (cand_metric[r] < best_any_m) { best_any = r; best_any_m = cand_metric[r]; }
    }
    const int k = best_crc >= 0 ? best_crc : best_any;
    if (k >= 0) { memcpy(payload, cand + 55 * (size_t)k, 55); *which = k; }
    return 0;
}

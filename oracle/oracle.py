"""ctypes front-end of oracle/liboracle.so (C restatement of the reference's hot path).

TEST INFRASTRUCTURE ONLY -- see oracle/__init__.py.  The C sources (oracle/c/*.c) cite the
reference lines they follow; tests/test_oracle_*.py pin them against vectors captured from the
reference itself (tests/golden/, generator oracle/refshim/gen_golden.py).
"""
from __future__ import annotations

import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(_HERE, "liboracle.so")
_lib = None


def build(force: bool = False) -> str:
    if force or not os.path.exists(LIB) or any(
            os.path.getmtime(os.path.join(_HERE, "c", f)) > os.path.getmtime(LIB)
            for f in os.listdir(os.path.join(_HERE, "c")) if f.endswith(".c")):
        subprocess.check_call(["make", "-C", os.path.join(_HERE, "c"), "-B"], stdout=subprocess.DEVNULL)
    return LIB


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(LIB)
        _lib.eso_cfar_threshold.restype = ctypes.c_double
        _lib.eso_sum_f32.restype = ctypes.c_float
        _lib.eso_crc8.restype = ctypes.c_uint8
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(ctypes.c_void_p)


def _c(a, dtype) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=dtype)


# ------------------------------------------------------------------------------- math
def exp_vec(x): x = _c(x, np.float64); y = np.empty_like(x); lib().eso_exp_vec(_p(x), _p(y), ctypes.c_int64(x.size)); return y
def log1p_vec(x): x = _c(x, np.float64); y = np.empty_like(x); lib().eso_log1p_vec(_p(x), _p(y), ctypes.c_int64(x.size)); return y
def logaddexp_vec(a, b):
    a = _c(a, np.float64); b = _c(b, np.float64); y = np.empty_like(a)
    lib().eso_logaddexp_vec(_p(a), _p(b), _p(y), ctypes.c_int64(a.size)); return y
def polar_f_vec(a, b):
    a = _c(a, np.float64); b = _c(b, np.float64); y = np.empty_like(a)
    lib().eso_polar_f_vec(_p(a), _p(b), _p(y), ctypes.c_int64(a.size)); return y
def penalty_vec(l, bit):
    l = _c(l, np.float64); y = np.empty_like(l)
    lib().eso_penalty_vec(_p(l), int(bit), _p(y), ctypes.c_int64(l.size)); return y
def sum_f32(a): a = _c(a, np.float32); return np.float32(lib().eso_sum_f32(_p(a), ctypes.c_int64(a.size)))


# ------------------------------------------------------------------------------- polar
_K = [448]                        # K of the code the polar functions below use (eso_polar_set_k); 448 unless a test says otherwise


class code_k:
    """`with oracle.code_k(512): ...` -- the polar functions inside use Polar(1024, 512)+CRC-8 (rtwm/fastpolar.py:209-234 takes any K)."""
    def __init__(self, K): self.K = int(K)
    def __enter__(self): self.old = lib().eso_polar_set_k(self.K); _K[0] = self.K; return self
    def __exit__(self, *exc): lib().eso_polar_set_k(self.old); _K[0] = self.old


def polar_tables():
    frozen = np.zeros(1024, np.uint8); dpos = np.zeros(_K[0], np.int32)
    lib().eso_polar_tables(_p(frozen), _p(dpos)); return frozen.astype(bool), dpos

def polar_encode(info_bits):
    info = _c(info_bits, np.uint8); code = np.zeros(1024, np.uint8)
    if info.size != _K[0] - 8:
        raise ValueError(f"need {_K[0] - 8} information bits")
    lib().eso_polar_encode(_p(info), _p(code)); return code

def crc8(bits): b = _c(bits, np.uint8); return int(lib().eso_crc8(_p(b), int(b.size)))

def polar_hard(llr):
    llr = _c(llr, np.float64); info = np.zeros(_K[0] - 8, np.uint8)
    ok = lib().eso_polar_hard(_p(llr), _p(info)); return info, bool(ok)

def scl_list(llr, L):
    """-> (n, info[L,K-8], metric[L], crc[L]) in ascending-metric (stable) order."""
    llr = _c(llr, np.float64)
    ci = np.zeros((L, _K[0] - 8), np.uint8); cm = np.zeros(L); cc = np.zeros(L, np.uint8)
    n = lib().eso_scl_list(_p(llr), int(L), _p(ci), _p(cm), _p(cc))
    return n, ci, cm, cc

def polar_decode(llr, L):
    """PolarCode.decode(llr, validator=None) -> (info bits[K-8], ok, took_list)."""
    llr = _c(llr, np.float64); info = np.zeros(_K[0] - 8, np.uint8); tl = ctypes.c_int(0)
    ok = lib().eso_polar_decode(_p(llr), int(L), _p(info), ctypes.byref(tl))
    return info, bool(ok), bool(tl.value)


# ------------------------------------------------------------------------------- DSP
def lfilter(b, a, x):
    b = _c(b, np.float64); a = _c(a, np.float64); x = _c(x, np.float32); y = np.empty(x.size, np.float64)
    lib().eso_lfilter(_p(b), _p(a), int(b.size), _p(x), _p(y), ctypes.c_int64(x.size), None); return y

def ncc(y, tpl):
    y = _c(y, np.float64); tpl = _c(tpl, np.float64); out = np.empty(max(0, y.size - tpl.size + 1), np.float64)
    lib().eso_ncc(_p(y), ctypes.c_int64(y.size), _p(tpl), int(tpl.size), _p(out)); return out

def cfar_threshold(corr):
    corr = _c(corr, np.float64); med = ctypes.c_double(); mad = ctypes.c_double()
    thr = lib().eso_cfar_threshold(_p(corr), ctypes.c_int64(corr.size), ctypes.byref(med), ctypes.byref(mad))
    return thr, med.value, mad.value

def pick_peaks(corr, thr, min_distance=607, max_peaks=32):
    corr = _c(corr, np.float64); pk = np.full(max_peaks, -1, np.int32); tot = ctypes.c_int(); fb = ctypes.c_int()
    n = lib().eso_pick_peaks(_p(corr), ctypes.c_int64(corr.size), ctypes.c_double(thr), int(min_distance), _p(pk),
                             int(max_peaks), ctypes.byref(tot), ctypes.byref(fb))
    return pk[:n].copy(), tot.value, bool(fb.value)

def llr(frame, pn_payload_bits, taps):
    """-> (llr float32[1024], best_s, best_score, second_score)."""
    frame = _c(frame, np.float64); pn = _c(pn_payload_bits, np.uint8); h = _c(taps, np.float32)
    if pn.size < 1024:
        raise ValueError("need 1024 PN bits")
    out = np.zeros(1024, np.float32); diag = np.zeros(4)
    lib().eso_llr(_p(frame), int(frame.size), _p(pn), _p(h), int(h.size), _p(out), _p(diag))
    return out, int(diag[0]), float(diag[1]), float(diag[2])


# ------------------------------------------------------------------------------- metric unit
def decode_frame(frame_f32, ba, tpl, taps, pn_full_bits, L=8, start=0):
    """sync + LLR(variant 0) + SCL-L for one record, as the reference would (validator None)."""
    y = lfilter(ba[:9], ba[9:], frame_f32)
    corr = ncc(y, tpl)
    thr, _, _ = cfar_threshold(corr)
    peaks, tot, fb = pick_peaks(corr, thr)
    if isinstance(start, str):                       # "peak": demodulate at the first detected peak (config-3 windows)
        start = int(peaks[0]) if tot else 0
    l, best_s, s0, s1 = llr(y[start:start + 1215], pn_full_bits[191:1215], taps)
    info, ok, took = polar_decode(l.astype(np.float64), L)
    return dict(y=y, corr=corr, thr=thr, peaks=peaks, npeaks=tot, fallback=fb, llr=l, best_s=best_s,
                info=np.packbits(info).tobytes(), ok=ok, took_list=took)


def decode_header(frame, hdr_pn_bits, taps):
    """WatermarkDetector._decode_header -> (ok, value, score, best_s)."""
    frame = _c(frame, np.float64); pn = _c(hdr_pn_bits, np.uint8); h = _c(taps, np.float32)
    out = np.zeros(3); diag = np.zeros(2)
    lib().eso_decode_header(_p(frame), int(frame.size), _p(pn), _p(h), int(h.size), _p(out), _p(diag))
    return bool(out[0]), int(out[1]), float(out[2]), int(diag[0])


# ------------------------------------------------------------------------------- AEAD validator (section 8 f-2)
def chacha20_block(key: bytes, counter: int, nonce: bytes) -> bytes:
    out = np.zeros(64, np.uint8)
    lib().eso_chacha20_block(_p(np.frombuffer(key, np.uint8)), ctypes.c_uint32(counter), _p(np.frombuffer(nonce, np.uint8)), _p(out))
    return out.tobytes()

def poly1305(otk: bytes, msg: bytes) -> bytes:
    tag = np.zeros(16, np.uint8); m = np.frombuffer(msg, np.uint8) if msg else np.zeros(1, np.uint8)
    lib().eso_poly1305(_p(np.frombuffer(otk, np.uint8)), _p(m), ctypes.c_size_t(len(msg)), _p(tag))
    return tag.tobytes()

def aead_open(key: bytes, nonce: bytes, aad: bytes, ct_and_tag: bytes):
    """RFC 8439 open -> plaintext bytes, or None when the tag does not verify."""
    ct, tag = ct_and_tag[:-16], ct_and_tag[-16:]
    out = np.zeros(max(1, len(ct)), np.uint8)
    a = np.frombuffer(aad, np.uint8) if aad else np.zeros(1, np.uint8)
    c = np.frombuffer(ct, np.uint8) if ct else np.zeros(1, np.uint8)
    ok = lib().eso_aead_open(_p(np.frombuffer(key, np.uint8)), _p(np.frombuffer(nonce, np.uint8)), _p(a), ctypes.c_size_t(len(aad)),
                             _p(c), ctypes.c_size_t(len(ct)), _p(np.frombuffer(tag, np.uint8)), _p(out))
    return out[:len(ct)].tobytes() if ok == 1 else None

def validate_blobs(key: bytes, blobs, ctrs):
    """Detector validator on [n,55] blobs with expected counters [n] -> (ok uint8[n], plain uint8[n,27])."""
    blobs = _c(blobs, np.uint8).reshape(-1, 55); ctrs = _c(ctrs, np.uint32)
    ok = np.zeros(len(blobs), np.uint8); plain = np.zeros((len(blobs), 27), np.uint8)
    k = np.frombuffer(key, np.uint8)
    for i in range(len(blobs)):
        ok[i] = lib().eso_validate_blob(_p(k), _p(blobs[i]), ctypes.c_uint32(int(ctrs[i])), _p(plain[i]))
    return ok, plain

def select_validated(key, ctr, hard, hard_ok, cand, cand_ok, cand_metric, n):
    """PolarCode.decode's candidate selection with the detector validator (key=None: validator None)
    -> (payload bytes, ok flag or -1, which)."""
    payload = np.zeros(55, np.uint8); which = ctypes.c_int(0)
    kp = _p(np.frombuffer(key, np.uint8)) if key is not None else None
    cand = _c(cand, np.uint8); cand_ok = _c(cand_ok, np.uint8); cand_metric = _c(cand_metric, np.float64)
    ok = lib().eso_select_validated(kp, ctypes.c_uint32(int(ctr)), _p(_c(hard, np.uint8)), int(hard_ok), _p(cand), _p(cand_ok),
                                    _p(cand_metric), int(n), _p(payload), ctypes.byref(which))
    return payload.tobytes(), int(ok), int(which.value)


# ------------------------------------------------------------------------------- key / PN / hop schedule (a18, f-3)
def aes128_encrypt(key: bytes, block: bytes) -> bytes:
    rk = np.zeros(176, np.uint8); out = np.zeros(16, np.uint8)
    lib().eso_aes128_expand(_p(np.frombuffer(key, np.uint8)), _p(rk))
    lib().eso_aes128_encrypt(_p(rk), _p(np.frombuffer(block, np.uint8)), _p(out)); return out.tobytes()

def hmac_sha256_short(key: bytes, msg: bytes) -> bytes:
    out = np.zeros(32, np.uint8)
    lib().eso_hmac_sha256_short(_p(np.frombuffer(key, np.uint8)), len(key), _p(np.frombuffer(msg, np.uint8)), len(msg), _p(out))
    return out.tobytes()

def schedule_rows(aes_key: bytes, band_key: bytes, ctrs):
    """-> (pn uint8 [n,152], band uint8 [n]) for 32-bit frame counters."""
    ctrs = np.asarray(ctrs, dtype=np.uint32); pn = np.zeros((ctrs.size, 152), np.uint8); band = np.zeros(ctrs.size, np.uint8)
    ak = np.frombuffer(aes_key, np.uint8); bk = np.frombuffer(band_key, np.uint8); b1 = np.zeros(1, np.uint8)
    for i, c in enumerate(ctrs):
        lib().eso_schedule_row(_p(ak), _p(bk), ctypes.c_uint32(int(c)), _p(pn[i]), _p(b1)); band[i] = b1[0]
    return pn, band


# ------------------------------------------------------------------------------- input conditioning (f-4)
def resample_plan(n_in: int, up: int, down: int, dtype):
    """SciPy's resample_poly set-up (Python in SciPy: filter design, padding, kept range), restated call for call:
    -> (h_trans_flip, h_per_phase, up, down, n_pre_remove, n_out, compute dtype) or None when up == down."""
    import math
    from scipy.signal import firwin
    g = math.gcd(int(up), int(down)); up = int(up) // g; down = int(down) // g
    if up == down == 1:
        return None
    n_out = n_in * up; n_out = n_out // down + bool(n_out % down)
    max_rate = max(up, down); half_len = 10 * max_rate
    h = firwin(2 * half_len + 1, 1.0 / max_rate, window=("kaiser", 5.0))
    if np.issubdtype(np.dtype(dtype), np.floating):
        h = h.astype(dtype)
    h *= up
    n_pre_pad = down - half_len % down; n_post_pad = 0; n_pre_remove = (half_len + n_pre_pad) // down
    def out_len(len_h):                                   # scipy.signal._upfirdn._output_len
        in_len_copy = n_in + (len_h + (-len_h % up)) // up - 1
        nt = in_len_copy * up
        need = nt // down + (1 if nt % down > 0 else 0)
        return need
    while out_len(len(h) + n_pre_pad + n_post_pad) < n_out + n_pre_remove:
        n_post_pad += 1
    h = np.concatenate((np.zeros(n_pre_pad, h.dtype), h, np.zeros(n_post_pad, h.dtype)))
    ctype = np.result_type(h.dtype, np.dtype(dtype), np.float32)
    h = np.asarray(h, ctype)
    padlen = len(h) + (-len(h) % up)
    hf = np.zeros(padlen, ctype); hf[:len(h)] = h
    h_tf = np.ascontiguousarray(hf.reshape(-1, up).T[:, ::-1].ravel())
    return h_tf, padlen // up, up, down, n_pre_remove, n_out, ctype

def resample_poly(x, up: int, down: int):
    """scipy.signal.resample_poly(x, up, down) for a 1-D signal through the C restatement of upfirdn's inner loop."""
    x = np.asarray(x)
    plan = resample_plan(x.shape[0], up, down, x.dtype)
    if plan is None:
        return x.copy()
    h_tf, hpp, up, down, y0, n_out, ctype = plan
    xc = _c(x, ctype); out = np.zeros(n_out, ctype)
    fn = lib().eso_upfirdn_f32 if ctype == np.float32 else lib().eso_upfirdn_f64
    fn(_p(xc), ctypes.c_long(xc.size), _p(h_tf), ctypes.c_long(hpp), ctypes.c_long(up), ctypes.c_long(down),
       ctypes.c_long(y0), ctypes.c_long(n_out), _p(out))
    return out

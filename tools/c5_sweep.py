"""BASELINE config 5 without the MP3 codec (none in the image): list size swept over 1/4/8/16.
(i)  whole path (sync + LLR + SCL-L) on 16 384 frames with AWGN: frames/s and the bit error rate of the decoded
     payload against the transmitted one (the reference's chain does not recover payloads -- SURVEY section 4 -- so this
     BER is ~0.5 by construction; it is reported because the config asks for it);
(ii) the list decoder alone on AWGN-corrupted polar codewords (BPSK, LLR = 2y/sigma^2): frame and bit error rates
     versus L, which is what the list size buys."""
import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from echoseal_amd.embedder import WatermarkEmbedder, synthetic_payloads
from echoseal_amd.utils import band_index
from echoseal_amd.engine import RxEngine
KEY = b"\xAA" * 32
eng = RxEngine(0, list_size_max=16); dev = eng.device
U, B = 1024, 16384
tx = WatermarkEmbedder(KEY); ctrs = list(range(U))
payloads = synthetic_payloads(tx.sec, ctrs)
frames = tx.make_frames(ctrs, payloads)
band = np.array([band_index(KEY, c) for c in ctrs], np.uint8); pn = tx.sec.pn_bytes_batch(ctrs, 152)
rng = np.random.default_rng(5)
noisy = (np.tile(frames, (B // U, 1)) + rng.normal(0, 0.05, (B, 1215))).astype(np.float32)
f = torch.from_numpy(noisy).to(dev); b = torch.from_numpy(band).to(dev).repeat(B // U); p = torch.from_numpy(pn).to(dev).repeat(B // U, 1)
want = torch.from_numpy(np.unpackbits(np.frombuffer(b"".join(payloads), np.uint8).reshape(U, 55), axis=1)).to(dev).repeat(B // U, 1)
print("(i) sync + LLR + SCL-L on 16 384 noisy frames")
for L in (1, 4, 8, 16):
    eng.decode_batch(f, b, p, list_size=L); torch.cuda.synchronize()
    t0 = time.perf_counter(); sy, llr, scl = eng.decode_batch(f, b, p, list_size=L); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    payload, ok, which = eng.select(scl)
    got = torch.from_numpy(np.unpackbits(payload.cpu().numpy(), axis=1)).to(dev)
    ber = float((got != want).float().mean())
    print(f"  L={L:2d}: {B / dt:10.0f} frames/s   payload BER {ber:.3f}   CRC-ok decodes {int((ok == 1).sum())}/{B}")
print("(ii) list decoder alone, AWGN on polar codewords (B = 16 384 per point)")
info = torch.from_numpy(rng.integers(0, 256, (B, 55), dtype=np.uint8)).to(dev)
code = eng.polar_encode(info).to(torch.float32)
ibits = torch.from_numpy(np.unpackbits(info.cpu().numpy(), axis=1)).to(dev)
for sigma in (0.30, 0.40, 0.50):
    g = torch.Generator(device=dev); g.manual_seed(int(sigma * 100))
    y = (2 * code - 1) + sigma * torch.randn(code.shape, device=dev, generator=g)
    llr = (2.0 / sigma ** 2) * y
    row = []
    for L in (1, 4, 8, 16):
        eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize()
        t0 = time.perf_counter(); res = eng.scl(llr, list_size=L, skip_if_hard_ok=False); torch.cuda.synchronize(); dt = time.perf_counter() - t0
        payload, ok, which = eng.select(res)
        got = torch.from_numpy(np.unpackbits(payload.cpu().numpy(), axis=1)).to(dev)
        fer = float((got != ibits).any(dim=1).float().mean()); ber = float((got != ibits).float().mean())
        row.append(f"L={L:2d} FER {fer:.4f} BER {ber:.5f} ({B / dt / 1e3:.0f} k/s)")
    print(f"  sigma {sigma:.2f}: " + " | ".join(row))

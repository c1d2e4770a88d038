# first GPU parity run: every kernel vs the C oracle on a handful of frames
import sys, time, numpy as np, torch
sys.path.insert(0, '.')
from echoseal_amd.engine import RxEngine
from echoseal_amd.embedder import WatermarkEmbedder, synthetic_payloads
from echoseal_amd.utils import band_index
from echoseal_amd.tables import pack_tables
from oracle import oracle as O
key=b"\xAA"*32
tx=WatermarkEmbedder(key)
B=int(sys.argv[1]) if len(sys.argv)>1 else 64
ctrs=list(range(B))
pl=synthetic_payloads(tx.sec,ctrs)
frames=tx.make_frames(ctrs,pl)
rng=np.random.default_rng(3)
frames[B//2:]+= rng.normal(0,0.2,frames[B//2:].shape).astype(np.float32)
band=np.array([band_index(key,c) for c in ctrs],np.uint8)
pn=tx.sec.pn_bytes_batch(ctrs,152)
ba,tpl,taps,ntaps,fz=pack_tables()
eng=RxEngine(0)
d=eng.device
t0=time.time()
sy,llr,scl=eng.decode_batch(torch.from_numpy(frames).to(d),torch.from_numpy(band).to(d),torch.from_numpy(pn).to(d),list_size=8,keep_corr=True)
torch.cuda.synchronize(); print("gpu pipeline time",time.time()-t0)
y=sy.y.cpu().numpy(); corr=sy.corr.cpu().numpy(); thr=sy.thr.cpu().numpy(); pk=sy.peaks.cpu().numpy(); npk=sy.npeaks.cpu().numpy()
llr_h=llr.cpu().numpy()
res=eng.scl(llr,list_size=8,skip_if_hard_ok=False)
ci=res.cand_info.cpu().numpy(); cm=res.cand_metric.cpu().numpy(); cok=res.cand_ok.cpu().numpy(); hi=res.hard_info.cpu().numpy(); hok=res.hard_ok.cpu().numpy()
bad=dict(y=0,corr=0,thr=0,peaks=0,llr=0,hard=0,cinfo=0,cmetric=0,cok=0)
for i in range(B):
    b=band[i]
    pnb=np.unpackbits(pn[i])[:1215]
    o=O.decode_frame(frames[i],ba[b],tpl[b],taps[b,:ntaps[b]],pnb,L=8)
    bad['y']+= not np.array_equal(o['y'],y[i]); bad['corr']+= not np.array_equal(o['corr'],corr[i]); bad['thr']+= o['thr']!=thr[i]
    n=npk[i]&0xffff; fb=bool(npk[i]>>30)
    bad['peaks']+= not (np.array_equal(o['peaks'][:n],pk[i,:n]) and fb==o['fallback'] and n==min(o['npeaks'],32))
    bad['llr']+= not np.array_equal(o['llr'],llr_h[i])
    hinfo,hk=O.polar_hard(llr_h[i].astype(np.float64))
    bad['hard']+= not (np.packbits(hinfo).tobytes()==hi[i].tobytes() and hk==bool(hok[i]))
    nn,oi,om,oc=O.scl_list(llr_h[i].astype(np.float64),8)
    bad['cinfo']+= not np.array_equal(np.packbits(oi,axis=1),ci[i]); bad['cmetric']+= not np.array_equal(om,cm[i]); bad['cok']+= not np.array_equal(oc,cok[i])
    if i<2: print(i,'peaks',pk[i,:n],o['peaks'],'thr',thr[i],'metric0',cm[i,0],om[0],'llr maxdiff',np.abs(o['llr']-llr_h[i]).max(), 'corr maxdiff', np.abs(o['corr']-corr[i]).max())
print("MISMATCH COUNTS over",B,"frames:",bad)
for L in (1,2,4,16,32):
    r=eng.scl(llr[:8],list_size=L,skip_if_hard_ok=False)
    m=0
    for i in range(8):
        nn,oi,om,oc=O.scl_list(llr_h[i].astype(np.float64),L)
        m+= not (np.array_equal(np.packbits(oi,axis=1),r.cand_info[i].cpu().numpy()) and np.array_equal(om,r.cand_metric[i].cpu().numpy()) and np.array_equal(oc,r.cand_ok[i].cpu().numpy()))
    print("L",L,"mismatches",m)
# timing
for Bt in (1024,):
    f=torch.from_numpy(np.tile(frames,(Bt//B+1,1))[:Bt]).to(d); bb=torch.from_numpy(np.tile(band,Bt//B+1)[:Bt]).to(d); pp=torch.from_numpy(np.tile(pn,(Bt//B+1,1))[:Bt]).to(d)
    for rep in range(2):
        torch.cuda.synchronize(); t0=time.time(); sy_=eng.sync(f,bb); torch.cuda.synchronize(); t1=time.time(); l_=eng.llr(sy_.y,bb,pp); torch.cuda.synchronize(); t2=time.time(); s_=eng.scl(l_,list_size=8); torch.cuda.synchronize(); t3=time.time()
        print(f"B={Bt} sync {1e3*(t1-t0):.2f} ms llr {1e3*(t2-t1):.2f} ms scl {1e3*(t3-t2):.2f} ms -> {Bt/(t3-t0):.0f} frames/s")

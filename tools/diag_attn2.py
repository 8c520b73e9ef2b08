import numpy as np, torch, sys, os
sys.path.insert(0,'/root/repo')
from lmdeploy_amd import _ffi
from oracle import tm_oracle as o
from tests.gpu_helpers import DevCache, dev, host, st
from tests.test_gpu_fullsize import _random_cache
tm=_ffi.load(); f16=np.float16
bits,Hq,Hkv=int(sys.argv[1]),32,8
rng = np.random.default_rng(4 * 100 + Hq)
B, layer = 64, 1
klen = rng.integers(1024, 2048, B).tolist(); klen[0], klen[1] = 2047, 1024
L = o.BlockLayout(2, Hkv, 128, 64, bits)
oc, tables, total = _random_cache(rng, L, klen)
q = rng.standard_normal((B, Hq * 128)).astype(f16)
dc = DevCache(L, total, tables); dc.upload(oc)
klen_d = dev(np.asarray(klen, np.int32))
out = torch.zeros((B, Hq * 128), dtype=torch.float16, device='cuda')
_ffi.check(tm.tm_decode_attention(out.data_ptr(), dev(q).data_ptr(), Hq * 128, klen_d.data_ptr(), B, Hq, 0.0, 1, None, dc.view(layer), st()))
got=host(out).reshape(B,Hq,128).astype(np.float32)
worst=[]
for b in range(B):
    Ks,Vs=[],[]
    for hd in range(Hkv):
        kd,vd=oc.load_dequant(tables[b],layer,hd,0,klen[b],'decode'); Ks.append(kd); Vs.append(vd)
    ref64=o.attention_reference_unfused(q[b].reshape(Hq,128),np.stack(Ks),np.stack(Vs))
    e=np.abs(got[b]-ref64)
    h=int(np.argmax(e.max(1)))
    # softmax peakedness of the worst head
    kv=h//(Hq//Hkv)
    s=(q[b].reshape(Hq,128)[h].astype(np.float64)@np.stack(Ks)[kv].astype(np.float64).T)/np.sqrt(128)
    p=np.exp(s-s.max()); 
    worst.append((e.max(), b, klen[b], h, int(np.argmax(e[h])), p.sum(), np.sort(p)[-3:].round(3).tolist(), int(np.argmax(s)), float(np.abs(ref64[h]).max())))
worst.sort(reverse=True)
for w in worst[:8]: print('err %.5f seq %d ctx %d head %d dim %d  l=%.2f top-p %s argmax_tok %d refmax %.3f' % w)
print('median err', np.median([w[0] for w in worst]))

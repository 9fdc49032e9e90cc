import numpy as np, collections, sys
d=np.load(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/dbg/phase.npy').astype(np.int64)
nb=d.shape[0]
T0,Ts,Te,Tx=d[:,:,0],d[:,:,1],d[:,:,2],d[:,:,3]
T=d[:,:,8:8+48].reshape(nb,4,8,6)
print('blocks',nb,'| prologue %.0f | loop %.0f | epilogue %.0f | total life %.0f'%((Ts-T0).mean(),(Te-Ts).mean(),(Tx-Te).mean(),(Tx-T0).mean()))
names=['issue global loads','first frag lgkm wait','MFMA phase (9 taps)','vmcnt(0) wait','ds_write+lgkm','barrier wait']
ph=np.zeros((nb,4,8,6))
for i in range(5): ph[...,i]=T[...,i+1]-T[...,i]
nxt=np.concatenate([T[:,:,1:,0],Te[:,:,None]],axis=2)
ph[...,5]=nxt-T[...,5]
for i,n in enumerate(names):
    v=ph[...,i]
    print('%-22s mean %7.0f p50 %7.0f p90 %7.0f max %7.0f'%(n,v.mean(),np.median(v),np.percentile(v,90),v.max()))
hw=d[:,0,5]; xcc=d[:,0,4]
cu=(hw>>8)&0xf; sh=(hw>>12)&1; se=(hw>>13)&0x7
key=xcc*1000+se*100+sh*20+cu
groups=collections.defaultdict(list)
for b in range(nb): groups[int(key[b])].append(b)
occ=[];gaps=[]
for cuid,bl in groups.items():
    st=d[bl,0,0]; en=d[bl,:,3].max(axis=1)
    t0=st.min(); span=en.max()-t0
    ev=sorted([(s,1) for s in st]+[(e,-1) for e in en]); cur=0; last=t0; hist=collections.Counter()
    for t,dl in ev: hist[cur]+=t-last; last=t; cur+=dl
    occ.append([hist[i]/span for i in range(4)])
    o=np.argsort(st); sst=st[o]; sen=np.sort(en)
    for i in range(2,len(bl)): gaps.append(sst[i]-sen[i-2])
print('CUs',len(groups),'frac time with 0/1/2/3 blocks resident',np.round(np.mean(occ,axis=0),3),' dispatch gap (end of block -> start of replacement): mean %.0f p50 %.0f'%(np.mean(gaps),np.median(gaps)))

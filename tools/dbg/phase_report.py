import numpy as np, collections, sys
d=np.load(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/dbg/phase.npy').astype(np.int64)   # [blk][wave][8]
nb=d.shape[0]
T0,Ts,Te,Tx=d[:,:,0],d[:,:,1],d[:,:,2],d[:,:,3]
print('blocks',nb,'| prologue %.0f | K loop %.0f | epilogue %.0f | total life %.0f  (cycles, mean per wave)'%((Ts-T0).mean(),(Te-Ts).mean(),(Tx-Te).mean(),(Tx-T0).mean()))
print('   p50/p90: prologue %d/%d  loop %d/%d  epilogue %d/%d'%(*np.percentile(Ts-T0,[50,90]),*np.percentile(Te-Ts,[50,90]),*np.percentile(Tx-Te,[50,90])))
hw=d[:,0,5]; xcc=d[:,0,4]
cu=(hw>>8)&0xf; sh=(hw>>12)&1; se=(hw>>13)&0x7
key=xcc*1000+se*100+sh*20+cu
groups=collections.defaultdict(list)
for b in range(nb): groups[int(key[b])].append(b)
occ=[];gaps=[];spans=[]
for cuid,bl in groups.items():
    st=d[bl,:,0].min(axis=1); en=d[bl,:,3].max(axis=1)
    t0=st.min(); span=en.max()-t0; spans.append(span)
    ev=sorted([(s,1) for s in st]+[(e,-1) for e in en]); cur=0; last=t0; hist=collections.Counter()
    for t,dl in ev: hist[cur]+=t-last; last=t; cur+=dl
    occ.append([hist[i]/span for i in range(4)])
    sst=np.sort(st); sen=np.sort(en)
    for i in range(2,len(bl)): gaps.append(sst[i]-sen[i-2])
print('CUs',len(groups),'span/CU %.0f'%np.mean(spans),'frac time with 0/1/2/3 blocks resident',np.round(np.mean(occ,axis=0),3),' gap block end -> replacement start: mean %.0f p50 %.0f'%(np.mean(gaps),np.median(gaps)))

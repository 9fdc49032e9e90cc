import numpy as np, sys
d=np.load(sys.argv[1] if len(sys.argv)>1 else 'gpurun_out/dbg/phase.npy').astype(np.int64)   # [blk][wave][8]
d=d[d[:,0,7]>0]
life=d[:,:,1]-d[:,:,0]; loop=d[:,:,2]; epi=d[:,:,3]; pro=d[:,:,6]; nt=d[:,:,7]
print('blocks',d.shape[0],'tiles/block',nt.mean(),'| per wave: life %.0f  prologue(first tile) %.0f  K-loops %.0f  epilogues %.0f  other %.0f'%(life.mean(),pro.mean(),loop.mean(),epi.mean(),(life-pro-loop-epi).mean()))
print('per tile: K loop %.0f (p50 %.0f p90 %.0f)   epilogue %.0f (p50 %.0f p90 %.0f)'%((loop/nt).mean(),np.median(loop/nt),np.percentile(loop/nt,90),(epi/nt).mean(),np.median(epi/nt),np.percentile(epi/nt,90)))

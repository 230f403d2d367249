import sys; sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import numpy as np, hudiff_oracle as ho, hudiff_amd
from hudiff_amd import synthetic as S
for kind in ('nb','ab'):
    cfg=dict(S.AB_CONFIG if kind=='ab' else S.NB_CONFIG); sd=S.random_state_dict(kind,cfg,seed=3)
    B=3; batch=S.synthetic_batch(kind,B,seed=5); tokens=batch['tokens'].copy()
    tokens[1]=np.where(tokens[1]==22,batch['truth'][1],tokens[1])
    m=(hudiff_amd.AntiTFNet if kind=='ab' else hudiff_amd.NanoAntiTFNet)(**cfg); m.load_state_dict(sd)
    for mode in ('off','faithful'):
        drop=ho.Dropout("philox", seed=99, rows=np.arange(B)+3, step=17) if mode=='faithful' else None
        c=cfg if mode=='faithful' else dict(cfg,dropout=0.0)
        o32=ho.OracleNet(kind,c,sd); o64=ho.OracleNet(kind,c,sd,dtype=np.float64)
        o32.trace={}; o64.trace={}
        l32=o32(tokens,batch['region'],batch['chain'],dropout=drop); l64=o64(tokens,batch['region'],batch['chain'],dropout=drop)
        m.debug_stop_after(0)
        got=m(tokens,batch['region'],batch['chain'],dropout=mode,seed=99,row0=3,step=17)
        print(kind,mode,'logits: hip-o64',np.abs(got-l64).max(),'o32-o64',np.abs(l32-l64).max())
        y=m.debug_read('Y',B); print('   final Y: hip-o64',np.abs(y-o64.trace['att4']).max(),'o32-o64',np.abs(o32.trace['att4']-o64.trace['att4']).max(), 'scale', np.abs(o64.trace['att4']).max())
        pos=m.debug_read('POS',B); print('   pos: hip-o64',np.abs(pos-o64.trace['pos']).max(),'o32-o64',np.abs(o32.trace['pos']-o64.trace['pos']).max())
        for st,name in ((1,'feature'),(2,'conv'),(3,'att0'),(4,'att1'),(5,'att2')):
            m.debug_stop_after(st); m(tokens,batch['region'],batch['chain'],dropout=mode,seed=99,row0=3,step=17)
            buf=m.debug_read('FEAT' if st==1 else 'Y',B)
            print('   ',name,'hip-o64',np.abs(buf-o64.trace[name]).max(),'o32-o64',np.abs(o32.trace[name]-o64.trace[name]).max(),'scale',np.abs(o64.trace[name]).max())
    m.close()

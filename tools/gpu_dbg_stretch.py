import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
from oracle_lib import Oracle
np.set_printoptions(precision=5,suppress=True,linewidth=220)
b=ModelBlob.load('bed_bathing_stretch'); o=Oracle(b)
states=np.load(sys.argv[1])
n=len(states)
s=Stepper(b,n)
rng=np.random.RandomState(7)
ref=states.copy()
for k in range(4):
    a=rng.uniform(-1,1,(n,b.act_dim)).astype(np.float32)
    s.set_state(ref); obs,rew,done,info=s.step_host(a); g=s.get_state()
    prev=ref.copy()
    for i in range(n):
        oo=o.step(ref[i],a[i])
        d=np.abs(b.view(g[i:i+1])['q'][0]-b.view(ref[i:i+1])['q'][0])
        if d.max()>2e-4:
            print('step',k,'env',i,'q diff',d.max(),'at',int(d.argmax()),'obs diff',np.abs(obs[i]-oo[0]).max(),'ncon',info[i,6],oo[3][6],'rows',info[i,7],oo[3][7], 'force',info[i,0],oo[3][0])
            print(' diff',d)
            np.save('gpurun_out/bed_bad_state_%d_%d.npy'%(k,i), prev[i]); np.save('gpurun_out/bed_bad_action_%d_%d.npy'%(k,i), a[i])

import sys, numpy as np
sys.path.insert(0,'tests'); sys.path.insert(0,'.')
from assistive_gym_amd.blob import ModelBlob
from assistive_gym_amd.libagx import Stepper
from assistive_gym_amd.vec_env import build_reset_pool
from oracle_lib import Oracle
np.set_printoptions(precision=4,suppress=True,linewidth=220)
b=ModelBlob.load('feeding_stretch'); o=Oracle(b)
n=16
states=build_reset_pool(b,n,5001)
s=Stepper(b,n)
rng=np.random.RandomState(7)
ref=states.copy()
for k in range(4):
    s.set_state(ref)
    a=rng.uniform(-1,1,(n,b.act_dim)).astype(np.float32)
    obs,rew,done,info=s.step_host(a)
    got=s.get_state()
    for i in range(n):
        r0=ref[i].copy()
        oo=o.step(ref[i],a[i])
        bad = (not np.isfinite(rew[i])) or (not np.isfinite(oo[1])) or bool(done[i])!=oo[2] or np.abs(obs[i]-oo[0]).max()>1e-3
        if bad:
            print('step',k,'env',i,'dev rew',rew[i],'done',done[i],'info',info[i],'| ora rew',oo[1],oo[2],oo[3])
            print(' q0 ',b.view(r0[None])['q'][0][:16]); print(' qd0',b.view(r0[None])['qd'][0][:16])
            print(' dev q',b.view(got[i:i+1])['q'][0][:16]); print(' ora q',b.view(ref[i:i+1])['q'][0][:16])
            print(' act',a[i])
            np.save('gpurun_out/stretch_bad_state.npy', r0); np.save('gpurun_out/stretch_bad_action.npy', a[i])
            sys.exit(0)
print('all fine')

import os
"""CPU check of the engine's shared dynamics header (host build) against the oracle."""
import numpy as np, ctypes as C, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import oracle as O
sh=C.CDLL('tests/shim/libhost_shim.so')
dp=C.POINTER(C.c_double)
sh.shim_car_rollout.restype=C.c_double; sh.shim_car_reward.restype=C.c_double
p=O.car_default_params(); track=O.load_track(); tx,ty,tw=track
rng=np.random.default_rng(0)
def d(a): return a.ctypes.data_as(dp)
maxerr=0
for i in range(40000):
    s=np.zeros(8); s[0:2]=rng.uniform(-50,50,2); s[2]=rng.uniform(-3.14,3.14); s[3]=rng.uniform(-6,35); s[4]=rng.uniform(-8,8); s[5]=rng.uniform(-2,2); s[6]=rng.uniform(-0.31,0.31)
    if i%5==0: s[3]=rng.uniform(-0.3,0.3)
    if i%50==0: s[3]=0.0
    if i%100==0: s[3:6]=0.0
    if i%11==0: s[5]=rng.uniform(-12,12)
    a=rng.uniform(-1,1,2)
    if i%7==0: a[1]=0.0
    if i%3==0: a[1]=-abs(a[1])
    ref=O.car_step(p,s,a)
    g=s.copy(); sh.shim_car_action_step(d(p),d(g),C.c_double(a[0]),C.c_double(a[1]))
    e=np.max(np.abs(g-ref)/(np.abs(ref)+1e-2))
    if e>maxerr: maxerr=e; worst=(s,a,g,ref)
print('single-step max rel err',maxerr)
if maxerr>1e-9: print(worst)
mr=0
for i in range(5000):
    s=np.zeros(8); s[0]=rng.uniform(-20,270); s[1]=rng.uniform(-170,160); s[3]=rng.uniform(-5,30); s[4]=rng.uniform(-20,20)
    e=O.OracleEnv('car',1,track=track); e.state=s
    r1=e.reward(); r2=sh.shim_car_reward(d(p),len(tx),d(tx),d(ty),d(tw),d(s))
    mr=max(mr,abs(r1-r2)/abs(r1))
print('reward max rel err',mr)
env=O.OracleEnv('car',1,track=track)
maxc=0
for i in range(300):
    ctrl=np.clip(rng.normal(0,1,(50,2))*np.array([0.25,0.32])+np.array([0,rng.uniform(-1.0,0.8)]),-1,1)
    e=env.copy(); c=0.0
    for t in range(50):
        e.step(ctrl[t]); c-=e.reward()
    s=env.state.copy(); c2=sh.shim_car_rollout(d(p),len(tx),d(tx),d(ty),d(tw),d(s),d(np.ascontiguousarray(ctrl)),50)
    maxc=max(maxc,abs(c2-c)/abs(c))
print('rollout max rel cost err',maxc)

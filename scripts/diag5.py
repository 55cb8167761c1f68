import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, picha_amd
from picha_amd import _native as N
from util import *
def opt(k,v): N.check(N.lib().picha_hip_set_option(k.encode(), v))
rng=np.random.default_rng(1)
for (sw,sh,dw,dh,fs) in [(64,48,32,24,0.7),(37,29,16,11,0.7),(64,48,32,24,1.0)]:
  img=rand_image(rng,sw,sh,'rgba')
  exp=oracle_resize(img,dw,dh,'cubic',fs)
  for pl in (1,2,4):
    opt('path',2); opt('per_lane',pl)
    try:
        got=picha_amd.resizeSync(img,{'width':dw,'height':dh,'filter':'cubic','filterScale':fs})
    except Exception as e:
        print(sw,dw,fs,'pl',pl,'ERR',e); continue
    a=got.pixels().astype(int); b=exp.pixels().astype(int)
    bad=np.abs(a-b).max(axis=2)
    ys,xs=np.nonzero(bad)
    print(sw,sh,dw,dh,fs,'pl',pl,N.lib().picha_hip_last_kernel().decode(),'maxdiff',bad.max(),'nbad',len(ys),'cols',sorted(set(xs.tolist()))[:20],'rows',sorted(set(ys.tolist()))[:12])

import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, picha_amd
from util import *
rng=np.random.default_rng(1)
for (sp,dp,w,h) in [('rgb','rgba',50,50),('grey','r16g16b16a16',50,50),('rgb','rgba',50,3),('rgb','rgba',48,3)]:
    img=rand_image(rng,w,h,sp)
    exp=oracle_convert(img,dp)
    got=picha_amd.colorConvertSync(img,{'pixel':dp})
    a,b=payload(got),payload(exp)
    bad=(a!=b)
    print(sp,dp,w,h,'mismatch',bad.sum(),'of',a.size,'rows bad:',np.nonzero(bad.any(axis=1))[0][:20].tolist())
    r=np.nonzero(bad.any(axis=1))[0]
    if len(r):
        y=r[0]; print(' row',y,'bad cols',np.nonzero(bad[y])[0][:40].tolist()); print(' got',a[y,:32]); print(' exp',b[y,:32])

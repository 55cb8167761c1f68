import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, picha_amd
from util import *
rng=np.random.default_rng(1)
for sp in PIXELS:
  for dp in PIXELS:
    for (w,h) in [(50,50),(16,2),(17,5),(1,1)]:
        img=rand_image(rng,w,h,sp)
        got=picha_amd.colorConvertSync(img,{'pixel':dp}); exp=oracle_convert(img,dp)
        a,b=payload(got),payload(exp)
        if not np.array_equal(a,b):
            idx=np.argwhere(a!=b)
            print(sp,dp,w,h,'mismatch',len(idx),'first',idx[:4].tolist(), a[tuple(idx[0])], b[tuple(idx[0])])
            break
print('done')

import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, picha_amd
from util import *
rng=np.random.default_rng(1)
img=rand_image(rng,50,50,'rgba')
exp=oracle_convert(img,'greya')
for i in range(4):
    got=picha_amd.colorConvertSync(img,{'pixel':'greya'})
    a,b=payload(got),payload(exp)
    print(i,'mismatch',(a!=b).sum(), a[0,:8], b[0,:8])
img=rand_image(rng,16,2,'rgba')
exp=oracle_convert(img,'rgba')
for i in range(3):
    got=picha_amd.colorConvertSync(img,{'pixel':'rgba'})
    a,b=payload(got),payload(exp)
    print(i,'copy mismatch',(a!=b).sum(), a[0,:8], b[0,:8])

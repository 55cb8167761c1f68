import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, picha_amd
from util import *
rng=np.random.default_rng(1)
bad=0; tot=0
for rep in range(3):
  for sp in PIXELS:
    for dp in PIXELS:
        img=rand_image(rng,50,50,sp)
        got=picha_amd.colorConvertSync(img,{'pixel':dp}); exp=oracle_convert(img,dp)
        tot+=1
        if not np.array_equal(payload(got),payload(exp)): bad+=1
print('bad',bad,'of',tot)

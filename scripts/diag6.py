import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, picha_amd
from picha_amd import _native as N
from util import *
rng=np.random.default_rng(1)
for (sw,sh,dw,dh,pix,f) in [(3840,2160,512,512,'grey','lanczos'),(3840,64,512,32,'grey','lanczos'),(384,64,51,32,'grey','lanczos'),(3840,64,512,32,'rgb','lanczos')]:
    img=rand_image(rng,sw,sh,pix)
    for path in (0,3):
        N.check(N.lib().picha_hip_set_option(b'path',path))
        try:
            got=picha_amd.resizeSync(img,{'width':dw,'height':dh,'filter':f})
            print(sw,sh,dw,dh,pix,'path',path,N.lib().picha_hip_last_kernel().decode())
        except Exception as e:
            print(sw,sh,dw,dh,pix,'path',path,'ERR',e)

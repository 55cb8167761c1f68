import sys; sys.path.insert(0,'.')
import numpy as np, picha_amd
from picha_amd import _native as N
from tests.util import rand_image
rng=np.random.default_rng(1)
N.check(N.lib().picha_hip_set_option(b"run",1))
for (sw,sh,dw,dh) in [(3840,264,480,33),(1920,300,320,50)]:
    big=rand_image(rng,sw+8,sh+3,"rgb",pad=4)
    for img in (big.subView(0,0,sw,sh), big.subView(8,3,sw,sh), big.subView(5,1,sw,sh)):
        for o in ({"width":dw,"height":dh,"filter":"lanczos"}, {"width":dw,"height":dh,"filter":"lanczos","filterScale":1.0}):
            print("==",sw,sh,dw,dh,o,img.stride,img.data.ctypes.data%4,flush=True)
            got=picha_amd.resizeSync(img,o)
            print(N.lib().picha_hip_last_kernel().decode(),flush=True)

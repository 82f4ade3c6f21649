import sys; sys.path.insert(0,'/root/repo')
import numpy as np, torch, frave_amd as fa
from tests.test_gpu_compact import _run
from tests.common import gen_image, random_params
ctx=fa.Context(0)
for (w,h,c) in ((16384,16384,1),(6000,4000,3)):
    P=fa.Plan(ctx,w,h,c); P.set_stream_order()
    vp,wp=random_params(11); params=np.stack([np.asarray(vp,np.float32).reshape(3,6),np.asarray(wp,np.float32).reshape(3,6)])
    img=gen_image("noise",w,h,c,5)
    for fit in (False,True):
        ref=_run(P,torch,[img],fit,params,compact=False)
        dr=_run(P,torch,[img],fit,params,compact=True,direct=True)
        ok=np.array_equal(dr[0],ref[0]) and np.array_equal(dr[1],ref[1]) and np.array_equal(dr[2],ref[2]) and np.array_equal(dr[3].view(np.uint32),ref[3].view(np.uint32))
        print(f"{w}x{h}x{c} fit={fit}: the scan's streams, histograms, counts and parameters == the int32 / gather route's: {ok}; symbols {ref[0].size}", flush=True)
        del ref, dr
    P.close()

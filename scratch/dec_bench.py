import sys,time,os
sys.path.insert(0,'classifier-pipeline_amd')
import numpy as np, torch
from cpx.engine import TrackEngine
from cpx.cptv import decode_clips_on_device, CptvReader
eng=TrackEngine(width=160,height=120,max_frames=1024)
paths=['tests/golden/possum.cptv','tests/golden/hedgehog.cptv']*64
t=time.time(); 
for p in paths[:8]: CptvReader(p).read_all()
th=time.time()-t
n8=sum(len(CptvReader(p).scan()[0]) for p in paths[:8])
print("host decode %.0f frames/s"%(n8/th))
decode_clips_on_device(eng,paths[:2])
t=time.time(); h,m,f,o=decode_clips_on_device(eng,paths); td=time.time()-t
print("device decode incl. inflate+index+upload: %d frames %.3f s -> %.0f frames/s"%(o[-1],td,o[-1]/td))
# kernel-only
import gzip
t=time.time()
for p in paths: gzip.open(p,'rb').read()
print("inflate only %.3f s"%(time.time()-t))
t=time.time()
for p in paths: CptvReader(p).scan()
print("inflate+scan %.3f s"%(time.time()-t))

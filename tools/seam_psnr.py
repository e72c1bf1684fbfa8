import sys
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, torch
from test_gpu_parity import _render_bands
from parity import psnr
from strolle_amd import scenes, CameraMode
size=(1920,1080)
single=_render_bands(torch, scenes.build_cornell, size, CameraMode.IMAGE, 0, 24, 1, 0)
for apron in (0, 8, 16, 32, 64, 128):
    tiled=_render_bands(torch, scenes.build_cornell, size, CameraMode.IMAGE, 0, 24, 4, apron)
    a=np.clip(tiled[...,:3],0,1); b=np.clip(single[...,:3],0,1)
    diff=np.abs(a-b).max(axis=2)
    print("apron", apron, "PSNR", round(psnr(a,b),2), "max abs", float(diff.max()), "rows>1e-3:", int((diff.max(axis=1)>1e-3).sum()))

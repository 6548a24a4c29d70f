import ctypes as C, sys
sys.path.insert(0, "cilrs-autonomous-driving-carla_amd")
import torch, torch.nn.functional as F
from cilrs_mi355 import _lib as L
lib = L.lib(); st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
N,H,W=1,88,200
g=torch.Generator().manual_seed(3)
x=torch.randn(N,3,H,W,generator=g); w=torch.randn(64,3,7,7,generator=g)/147**0.5
ref=F.conv2d(x.double(),w.double(),None,2,3)
x4=torch.zeros(N,H,W,4); x4[...,:3]=x.permute(0,2,3,1); x4=x4.cuda()
y=torch.full((N,44,100,64),float("nan"),device="cuda")
L.check(lib.cilrs_stem_conv_fwd(L.ptr(x4),L.ptr(w.permute(0,2,3,1).contiguous().cuda()),L.ptr(y),None,N,H,W,None,st))
torch.cuda.synchronize()
d=(y.cpu().permute(0,3,1,2).double()-ref).abs()
print("max",d.max().item(),"nan",torch.isnan(d).sum().item())
bad=(d>1e-4)
print("bad frac",bad.float().mean().item())
print("bad by channel", bad.sum((0,2,3))[:8].tolist(), bad.sum((0,2,3))[32:40].tolist())
print("bad by row", bad.sum((0,1,3))[:12].tolist())
print("bad by col", bad.sum((0,1,2))[:16].tolist())
# single-tap weights to find which tap is wrong
for (kh,kw,c) in ((0,0,0),(0,1,0),(0,6,2),(3,3,1),(6,5,0),(6,6,2)):
    w2=torch.zeros(64,3,7,7); w2[:,c,kh,kw]=1.0
    ref2=F.conv2d(x.double(),w2.double(),None,2,3)
    L.check(lib.cilrs_stem_conv_fwd(L.ptr(x4),L.ptr(w2.permute(0,2,3,1).contiguous().cuda()),L.ptr(y),None,N,H,W,None,st))
    torch.cuda.synchronize()
    d2=(y.cpu().permute(0,3,1,2).double()-ref2).abs()
    print("tap",kh,kw,c,"max diff",d2.max().item())

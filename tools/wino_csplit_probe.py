import sys, ctypes as C
sys.path.insert(0, "cilrs-autonomous-driving-carla_amd")
import torch
from cilrs_mi355 import _lib as L
lib = L.lib()
N,H,W,Cc=128,6,13,256
x=torch.randn(N,H,W,Cc,device="cuda"); U=torch.randn(16*Cc*Cc,device="cuda"); y=torch.empty(N,H,W,Cc,device="cuda")
part=torch.empty(2*Cc*1024,device="cuda"); slabs=torch.empty(4*N*H*W*Cc,device="cuda")
cs,rows=C.c_int(0),C.c_int(0)
st=C.c_void_p(torch.cuda.current_stream().cuda_stream)
L.check(lib.cilrs_conv2d_wino_split(L.ptr(x),L.ptr(U),L.ptr(y),None,L.ptr(part),N,H,W,Cc,Cc,L.ptr(slabs),slabs.numel(),C.byref(cs),C.byref(rows),st))
torch.cuda.synchronize(); print("csplit",cs.value,"rows",rows.value, "cus", torch.cuda.get_device_properties(0).multi_processor_count)

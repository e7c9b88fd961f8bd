import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check
lib=_lib.load(); s=torch.cuda.current_stream().cuda_stream
N,H,W,ci,co=8,180,240,256,128
M=N*H*W
def run(tag, x, w, n=6):
    y=torch.empty(M,co,device='cuda'); b=torch.zeros(co,device='cuda'); P=(M+63)//64; st=torch.empty(2*P*co,device='cuda')
    f=lambda: check(lib.cvk_conv3x3_fwd(x.data_ptr(),w.data_ptr(),b.data_ptr(),y.data_ptr(),st.data_ptr(),N,H,W,ci,co,co,s))
    f(); torch.cuda.synchronize()
    e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize()
    t=e0.elapsed_time(e1)/n*1e-3
    print(f"{tag:30s} {t*1e6:8.1f}us {18.0*M*ci*co/t/1e12:6.1f} TF", flush=True)
x=torch.randn(M,ci,device='cuda'); w=torch.randn(co,9*ci,device='cuda')*0.05
run("randn", x, w)
run("relu(randn)", torch.relu(x), w)
run("zeros", torch.zeros_like(x), w)
run("randn x, tiny w 1e-3", x, w*0.02)
run("randn long burst n=60", x, w, n=60)
# spread allocations: hold 40 GB of other tensors
hold=[torch.empty(1<<28,device='cuda') for _ in range(20)]
x2=torch.randn(M,ci,device='cuda')
run("after 20GB of allocations", x2, w)
# sustained load: run 300 iterations (about 0.5 s) then measure
for _ in range(3): run("sustained n=200", x, w, n=200)

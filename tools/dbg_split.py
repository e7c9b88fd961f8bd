import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check
lib=_lib.load(); s=torch.cuda.current_stream().cuda_stream
def one(N,H,W,ci,co):
    M=N*H*W; torch.manual_seed(1)
    x=torch.randn(M,ci,device='cuda'); w=torch.randn(co,9*ci,device='cuda')*0.05; b=torch.randn(co,device='cuda')
    ldy=(co+3)//4*4; P=(M+63)//64
    # fp64 reference through torch conv on CPU-free path: use double matmul on GPU via unfold is heavy; use fp64 direct sum with F.conv2d double
    xd=x.double().view(N,H,W,ci).permute(0,3,1,2); wd=w.double().view(co,3,3,ci).permute(0,3,1,2)
    ref=torch.nn.functional.conv2d(xd,wd,b.double(),padding=1).permute(0,2,3,1).reshape(M,co)
    outs={}
    for name,fn in (("fp32",lambda y,st: lib.cvk_conv3x3_fwd(x.data_ptr(),w.data_ptr(),b.data_ptr(),y.data_ptr(),st.data_ptr(),N,H,W,ci,co,ldy,s)),
                    ("split16",lambda y,st: lib.cvk_conv3x3_fwd_split(x.data_ptr(),w.data_ptr(),b.data_ptr(),y.data_ptr(),st.data_ptr(),N,H,W,ci,co,ldy,0,s)),
                    ("split32",lambda y,st: lib.cvk_conv3x3_fwd_split(x.data_ptr(),w.data_ptr(),b.data_ptr(),y.data_ptr(),st.data_ptr(),N,H,W,ci,co,ldy,1,s)),
                    ("bf16",lambda y,st: lib.cvk_conv3x3_fwd_bf16(x.data_ptr(),w.data_ptr(),b.data_ptr(),y.data_ptr(),st.data_ptr(),N,H,W,ci,co,ldy,s))):
        y=torch.zeros(M,ldy,device='cuda'); st=torch.zeros(2*P*co,device='cuda'); check(fn(y,st)); torch.cuda.synchronize()
        e=(y[:,:co].double()-ref); outs[name]=(e.abs().max().item(), (e.norm()/ref.norm()).item())
    print(f"N{N} {H}x{W} {ci}->{co}: " + "  ".join(f"{k}: max {v[0]:.2e} relL2 {v[1]:.2e}" for k,v in outs.items()))
for shp in [(1,5,7,64,128),(2,11,15,32,96),(1,22,30,128,64),(1,45,60,64,64),(2,9,4,96,40),(1,12,12,1024,128)]:
    one(*shp)

import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pytorch_camvid_amd import _lib
from pytorch_camvid_amd._lib import check
lib=_lib.load(); s=torch.cuda.current_stream().cuda_stream
def one(N,H,W,ci,co):
    M=N*H*W; torch.manual_seed(1)
    x=torch.randn(M,ci,device='cuda'); w=torch.randn(co,9*ci,device='cuda')*0.05; b=torch.randn(co,device='cuda')
    ldy=(co+3)//4*4; P=(M+63)//64
    y1=torch.zeros(M,ldy,device='cuda'); y2=torch.zeros(M,ldy,device='cuda'); s1=torch.zeros(2*P*co,device='cuda'); s2=torch.zeros(2*P*co,device='cuda')
    # reference: same kernel semantics on bf16-rounded operands in fp32
    xb=x.bfloat16().float(); wb=w.bfloat16().float()
    check(lib.cvk_conv3x3_fwd(xb.data_ptr(),wb.data_ptr(),b.data_ptr(),y1.data_ptr(),s1.data_ptr(),N,H,W,ci,co,ldy,s))
    check(lib.cvk_conv3x3_fwd_bf16(x.data_ptr(),w.data_ptr(),b.data_ptr(),y2.data_ptr(),s2.data_ptr(),N,H,W,ci,co,ldy,s))
    torch.cuda.synchronize()
    print(f"N{N} {H}x{W} {ci}->{co}: max|dy|={(y1-y2).abs().max().item():.3e} (|y|max {y1.abs().max().item():.2f}) stats diff {(s1-s2).abs().max().item():.3e}")
for shp in [(1,2,3,512,512),(1,5,7,64,128),(2,11,15,32,96),(1,22,30,128,12),(1,45,60,64,64),(2,9,4,96,40)]:
    one(*shp)

def wg(N,H,W,ci,co):
    M=N*H*W; torch.manual_seed(2)
    x=torch.randn(M,ci,device='cuda'); ldy=(co+3)//4*4; dy=torch.zeros(M,ldy,device='cuda'); dy[:,:co]=torch.randn(M,co,device='cuda')
    d1=torch.zeros(co,9*ci,device='cuda'); d2=torch.zeros(co,9*ci,device='cuda')
    xb=x.bfloat16().float(); dyb=dy.bfloat16().float()
    wsb=lib.cvk_conv3x3_wgrad_workspace_bytes(N,H,W,ci,co); ws=torch.zeros(wsb,dtype=torch.uint8,device='cuda')
    check(lib.cvk_conv3x3_wgrad(xb.data_ptr(),dyb.data_ptr(),d1.data_ptr(),N,H,W,ci,ci,co,ldy,ws.data_ptr(),wsb,s))
    wsb=lib.cvk_conv3x3_wgrad_bf16_workspace_bytes(N,H,W,ci,co); ws=torch.zeros(wsb,dtype=torch.uint8,device='cuda')
    check(lib.cvk_conv3x3_wgrad_bf16(x.data_ptr(),dy.data_ptr(),d2.data_ptr(),N,H,W,ci,ci,co,ldy,ws.data_ptr(),wsb,s))
    torch.cuda.synchronize()
    print(f"WGRAD N{N} {H}x{W} {ci}->{co}: max|d|={(d1-d2).abs().max().item():.3e} (|dw|max {d1.abs().max().item():.2f})")
for shp in [(2,3,1,64,64),(1,2,3,64,64),(1,5,7,128,128),(2,11,15,32,96),(1,22,30,128,64),(2,45,60,64,128),(1,1,1,64,64),(2,9,4,4,64),(8,90,120,256,256)]:
    wg(*shp)

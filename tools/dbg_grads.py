import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pytorch_camvid_amd as A
from pytorch_camvid_amd.modules import runner_of
torch.manual_seed(0)
net=A.UNet(3,12).cuda().train()
x=torch.randn(2,3,64,96,device='cuda'); t=torch.randint(0,12,(2,64,96),device='cuda')
A.CrossEntropyLoss()(net(x),t).backward()
f=runner_of(net)._flat[0]; lo,hi=f.data_ptr(), f.data_ptr()+4*f.numel()
bad=[(k,tuple(p.shape),p.stride(),p.grad.stride()) for k,p in net.named_parameters() if not (lo<=p.grad.data_ptr()<hi)]
print(len(bad), "grads were copied (not views of the flat buffer)")
for b in bad[:25]: print(b)

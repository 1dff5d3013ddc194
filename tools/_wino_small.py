import sys; sys.path.insert(0,'/root/repo')
import torch
from gddim_amd import ops
torch.manual_seed(0)
dev='cuda'
for (B,H,W,Cin,Cout) in [(2,8,8,32,64),(3,16,16,64,128),(1,32,32,32,68),(5,8,16,96,64)]:
    x=torch.randn(B,H,W,Cin,device=dev); w=torch.randn(Cout,9*Cin,device=dev)/(9*Cin)**.5
    b=torch.randn(Cout,device=dev); r=torch.randn(B,H,W,Cout,device=dev)
    import os
    os.environ['GDDIM_WINO_MINBLOCKS']='1'
    got=ops.conv3x3_winograd(x,w,b,r,alpha=1.3,out_scale=0.5)
    wt=w.double().view(Cout,3,3,Cin).permute(0,3,1,2)
    ref=(1.3*torch.nn.functional.conv2d(x.double().permute(0,3,1,2).cpu(), wt.cpu(), padding=1).permute(0,2,3,1)+b.double().cpu()+r.double().cpu())*0.5
    d=ops.conv2d(x,w,b,r,alpha=1.3,out_scale=0.5)
    print((B,H,W,Cin,Cout),'wino err',float((got.double().cpu()-ref).abs().max()),'direct err',float((d.double().cpu()-ref).abs().max()))

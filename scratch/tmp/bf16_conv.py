
import sys
sys.path.insert(0, '/root/repo')
import torch, torch.nn.functional as F
from pemp_amd import ops
dev = torch.device('cuda:0')
torch.manual_seed(0)
for (N,HW,Cin,Cout,k,dil,res) in ((2,21,64,64,1,1,False),(2,21,128,256,3,2,False),(2,33,256,128,1,1,True),(2,51,256,256,3,6,False)):
    x = torch.randn(N,HW,HW,Cin,device=dev)
    w = torch.randn(Cout,Cin,k,k,device=dev)*(1.0/(Cin*k*k)**0.5)
    b = torch.randn(Cout,device=dev)
    xb, wb = x.to(torch.bfloat16), w.to(torch.bfloat16)
    r = torch.randn(N,HW,HW,Cout,device=dev).to(torch.bfloat16) if res else None
    pad = dil if k==3 else 0
    ref = F.conv2d(xb.float().permute(0,3,1,2), wb.float(), b, 1, pad, dil).permute(0,2,3,1)
    if res: ref = ref + r.float()
    ref = F.relu(ref)
    packed = wb.permute(0,2,3,1).reshape(Cout,-1).contiguous()
    p = ops.ConvParams(packed, None, b, Cin, Cout, k, k, 1, pad, dil, packed.shape[1], False, True)
    for tile in (23,22,25,21,24,26,27):
        if Cout % ops.TILE_VARIANTS[tile][1]: continue
        y = ops.conv2d(xb, p, residual=r, tile=tile)
        err = (y.float()-ref).abs().max().item()
        y32 = ops.conv2d(xb, p, out=torch.empty(N,HW,HW,Cout,device=dev), tile=tile) if not res else None
        e32 = (y32-ref).abs().max().item() if y32 is not None else -1
        print((N,HW,Cin,Cout,k,dil,res), 'tile', tile, 'bf16-out err', round(err,4), 'f32-out err', round(e32,6), 'ref max', round(ref.abs().max().item(),2))

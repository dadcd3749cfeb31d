import torch, time
torch.manual_seed(0)
npad=65536
shapes=[(10,196),(3,12),(64,196),(64,64),(128,360),(128,130),(120,138),(64,122),(64,130),(64,66),(2,66),(24,130),(96,98),(6,96),(96,98),(3,96)]
ys=torch.randn(965,npad,device='cuda'); xs=torch.randn(2074,npad,device='cuda')
def run(mode,S=64):
    yr=0;xr=0;outs=[]
    for no,ns in shapes:
        g=ys[yr:yr+no]; x=xs[xr:xr+ns]; yr+=no; xr+=ns
        if mode=='mm': outs.append(g@x.t())
        else:
            per=npad//S
            outs.append(torch.bmm(g.view(no,S,per).transpose(0,1), x.view(ns,S,per).permute(1,2,0)).sum(0))
    return outs
for mode,S in (('mm',0),('bmm',64),('bmm',16),('bmm',256)):
    for _ in range(3): run(mode,S)
    torch.cuda.synchronize(); t=time.perf_counter()
    for _ in range(10): run(mode,S)
    torch.cuda.synchronize(); print(mode,S,(time.perf_counter()-t)/10*1e3,'ms per block (16 layers)')

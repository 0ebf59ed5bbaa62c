import sys, torch
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/10*1e3
for (N,H,Cin,Cout,K,s) in ((2048,9,64,64,3,1),(2048,20,32,64,4,2),(2048,23,64,64,3,1),(2048,49,32,64,4,2)):
    OH = (H-K)//s+1
    for dyt in (torch.bfloat16, torch.float32):
        dy = torch.randn(N,OH,OH,Cout,device=dev).to(dyt)
        act = torch.relu(torch.randn(N,H,H,Cin,device=dev)).to(torch.bfloat16)
        wt = (torch.randn(Cin,K,K,Cout,device=dev)*0.05).to(torch.bfloat16)
        dx = torch.empty(N,H,H,Cin,device=dev,dtype=torch.bfloat16)
        pos = (act.float()>0).reshape(-1,Cin//32,32).to(torch.int64)
        w = (pos << torch.arange(32,device=dev)).sum(-1); bits = torch.where(w>=2**31, w-2**32, w).to(torch.int32).t().contiguous().reshape(-1)
        t0 = timeit(lambda: kn.conv2d_bwd_data(dy,wt,dx,act,N,H,H,Cin,Cout,K,K,s,compute=kn.BF16))
        t1 = timeit(lambda: kn.conv2d_bwd_data(dy,wt,dx,act,N,H,H,Cin,Cout,K,K,s,compute=kn.BF16,relu_bits=bits))
        print(f"dgrad N={N} H={H} Cin={Cin} dy={dyt}: activation mask {t0:.1f} us, sign planes {t1:.1f} us")

# cold-cache variant of the gripper conv3 data gradient (the in-step situation): a 1 GB fill between launches
N,H,Cin,Cout,K,s = 2048,9,64,64,3,1
OH = 7
dy = torch.randn(N,OH,OH,Cout,device=dev).to(torch.bfloat16)
act = torch.relu(torch.randn(N,H,H,Cin,device=dev)).to(torch.bfloat16)
wt = (torch.randn(Cin,K,K,Cout,device=dev)*0.05).to(torch.bfloat16)
dx = torch.empty(N,H,H,Cin,device=dev,dtype=torch.bfloat16)
pos = (act.float()>0).reshape(-1,Cin//32,32).to(torch.int64)
w = (pos << torch.arange(32,device=dev)).sum(-1); bits = torch.where(w>=2**31, w-2**32, w).to(torch.int32).t().contiguous().reshape(-1)
junk = torch.empty(1 << 28, device=dev)
for name, kw in (("activation mask", {}), ("sign planes", {"relu_bits": bits})):
    ts = []
    for _ in range(6):
        junk.fill_(1.0)
        e0,e1 = torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record(); kn.conv2d_bwd_data(dy,wt,dx,act,N,H,H,Cin,Cout,K,K,s,compute=kn.BF16,**kw); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)*1e3)
    print(f"dgrad cold 9x9 {name}: " + " ".join(f"{t:.0f}" for t in ts) + " us")

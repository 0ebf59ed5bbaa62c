import sys, torch, os
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
L = [("static2",1024,49,49,32,64,4,2),("static3",1024,23,23,64,64,3,1),("grip2",1024,20,20,32,64,4,2),("grip3",1024,9,9,64,64,3,1)]
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/5
for name,N,H,W,Cin,Cout,K,s in L:
    OH,OW = kn.conv_out_hw(H,W,K,K,s)
    x = torch.randn(N,H,W,Cin,device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout,Cin,K,K,device=dev)/ (Cin*K*K)**0.5)
    w2d = w.permute(0,2,3,1).reshape(Cout,-1).contiguous().to(torch.bfloat16)
    wt = w.permute(1,2,3,0).contiguous().to(torch.bfloat16)
    b = torch.zeros(Cout,device=dev)
    y = torch.empty(N,OH,OW,Cout,device=dev,dtype=torch.bfloat16)
    dy = torch.randn(N,OH,OW,Cout,device=dev).to(torch.bfloat16)
    dx = torch.empty(N,H,W,Cin,device=dev,dtype=torch.bfloat16)
    flops = 2.0*N*OH*OW*Cout*Cin*K*K
    tf = timeit(lambda: kn.conv2d_fwd(x,w2d,b,y,N,H,W,Cin,Cout,K,K,s,False))
    td = timeit(lambda: kn.conv2d_bwd_data(dy,wt,dx,x,N,H,W,Cin,Cout,K,K,s))
    print(f"{name:8s} band={'off' if os.environ.get('HULC_NO_BAND') else 'on '} fwd {tf:.3f} ms ({flops/tf/1e9:.0f} TF/s)  dgrad {td:.3f} ms ({flops/td/1e9:.0f} TF/s)")

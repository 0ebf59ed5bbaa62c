import sys, torch, os
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/10
H,W,Cin,Cout,K,s = 23,23,64,64,3,1
for N in (256, 512, 1024, 2048, 4096):
    OH,OW = kn.conv_out_hw(H,W,K,K,s)
    x = torch.randn(N,H,W,Cin,device=dev).to(torch.bfloat16)
    w = (torch.randn(Cout,Cin,K,K,device=dev)/ (Cin*K*K)**0.5)
    w2d = w.permute(0,2,3,1).reshape(Cout,-1).contiguous().to(torch.bfloat16)
    b = torch.zeros(Cout,device=dev)
    y = torch.empty(N,OH,OW,Cout,device=dev,dtype=torch.bfloat16)
    tf = timeit(lambda: kn.conv2d_fwd(x,w2d,b,y,N,H,W,Cin,Cout,K,K,s,False))
    print(f"dbg={os.environ.get('HULC_BAND_DBG','0')} N={N:5d} fwd {tf*1e3:.1f} us")

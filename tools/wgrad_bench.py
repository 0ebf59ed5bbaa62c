import sys, torch, os
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
L = [("static1",1024,200,200,3,32,8,4,True),("static2",1024,49,49,32,64,4,2,False),("static3",1024,23,23,64,64,3,1,False),
     ("grip1",1024,84,84,3,32,8,4,True),("grip2",1024,20,20,32,64,4,2,False),("grip3",1024,9,9,64,64,3,1,False)]
for name,N,H,W,Cin,Cout,K,s,nchw in L:
    OH,OW = kn.conv_out_hw(H,W,K,K,s)
    x = torch.randn(N,Cin,H,W,device=dev) if nchw else torch.randn(N,H,W,Cin,device=dev).to(torch.bfloat16)
    dy = torch.randn(N,OH,OW,Cout,device=dev).to(torch.bfloat16)
    dw = torch.empty(Cout,Cin*K*K,device=dev); db = torch.empty(Cout,device=dev)
    for pf in ("0","1"):
        os.environ["HULC_WGRAD_PAIR_FASTEST"] = pf
        for _ in range(2): kn.conv2d_bwd_weight(x,dy,dw,db,N,H,W,Cin,Cout,K,K,s,nchw)
        torch.cuda.synchronize()
        e0,e1 = torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): kn.conv2d_bwd_weight(x,dy,dw,db,N,H,W,Cin,Cout,K,K,s,nchw)
        e1.record(); torch.cuda.synchronize()
        print(f"{name:8s} pair_fastest={pf}  {e0.elapsed_time(e1)/5:.3f} ms")

import sys, torch, os
sys.path.insert(0, '.')
from hulc2_amd import kernels as kn
dev = torch.device('cuda')
def timeit(fn):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    e0,e1 = torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1)/5
for name,N,H in (("static1",1024,200),("static1x2",2048,200),("grip1",1024,84),("grip1x2",2048,84)):
    OH = (H-8)//4+1
    x = torch.randn(N,3,H,H,device=dev)
    w2d = (torch.randn(32,192,device=dev)/192**0.5).to(torch.bfloat16)
    b = torch.zeros(32,device=dev)
    y = torch.empty(N,OH,OH,32,device=dev,dtype=torch.bfloat16)
    t = timeit(lambda: kn.conv2d_fwd(x,w2d,b,y,N,H,H,3,32,8,8,4,True,relu=True))
    byts = x.numel()*4 + y.numel()*2
    print(f"{name:10s} conv1 fwd {t:.3f} ms  {byts/t/1e9:.2f} TB/s (algorithmic bytes)  band={'off' if os.environ.get('HULC_NO_BAND_CONV1') else 'on'}")
# uint8 NHWC frames through the same entry point
for name,N,H in (("static1 u8",1024,200),("grip1 u8",1024,84)):
    OH = (H-8)//4+1
    x = torch.randint(0,256,(N,H,H,3),device=dev,dtype=torch.uint8)
    sh = torch.randint(0,9,(N,2),device=dev,dtype=torch.int32)
    w2d = (torch.randn(32,192,device=dev)/192**0.5).to(torch.bfloat16)
    b = torch.zeros(32,device=dev)
    y = torch.empty(N,OH,OH,32,device=dev,dtype=torch.bfloat16)
    t = timeit(lambda: kn.conv2d_fwd(x,w2d,b,y,N,H,H,3,32,8,8,4,True,relu=True,aug_shift=sh,aug_pad=4))
    byts = x.numel() + y.numel()*2
    print(f"{name:10s} conv1 fwd {t:.3f} ms  {byts/t/1e9:.2f} TB/s (algorithmic bytes)")
    dw = torch.empty(32,192,device=dev); db = torch.empty(32,device=dev)
    dy = torch.randn(N,OH,OH,32,device=dev).to(torch.bfloat16)
    t = timeit(lambda: kn.conv2d_bwd_weight(x,dy,dw,db,N,H,H,3,32,8,8,4,True,dw_oihw=True,aug_shift=sh,aug_pad=4))
    print(f"{name:10s} conv1 wgrad {t:.3f} ms  {(x.numel()+dy.numel()*2)/t/1e9:.2f} TB/s")
for name,N,H in (("static1 f32",1024,200),):
    OH = (H-8)//4+1
    x = torch.randn(N,3,H,H,device=dev)
    dw = torch.empty(32,192,device=dev); db = torch.empty(32,device=dev)
    dy = torch.randn(N,OH,OH,32,device=dev).to(torch.bfloat16)
    t = timeit(lambda: kn.conv2d_bwd_weight(x,dy,dw,db,N,H,H,3,32,8,8,4,True,dw_oihw=True))
    print(f"{name:10s} conv1 wgrad {t:.3f} ms  {(x.numel()*4+dy.numel()*2)/t/1e9:.2f} TB/s")

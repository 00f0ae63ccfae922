import sys; sys.path.insert(0,'/root/repo')
import torch, numpy as np, streamvbyte_amd as svb
n=1<<28
codec=svb.DeviceCodec(n)
g=torch.Generator(device='cuda'); g.manual_seed(9)
gr=torch.round(torch.randn(n,generator=g,device='cuda')*0.8).to(torch.int32)
out=torch.empty(svb.streamvbyte_max_compressedbytes(n),dtype=torch.uint8,device='cuda')
dec=torch.empty(n,dtype=torch.int32,device='cuda')
def t(f):
    for _ in range(2): f()
    ts=[]
    for _ in range(7):
        a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b))
    return float(np.median(ts))
for variant,flags,name in ((1,2,'0124 zz'),(0,2,'1234 zz'),(1,0,'0124 plain on |g|')):
    src = gr if flags else gr.abs()
    _,sz=codec.encode(src,out=out,variant=variant,flags=flags); size=int(sz.item())
    d=(size-(n+3)//4)/n
    te=t(lambda: codec.encode(src,out=out,variant=variant,flags=flags))
    s=out[:size+16].clone()
    td=t(lambda: codec.decode(s,n,out=dec,variant=variant,flags=flags))
    assert torch.equal(dec,src)
    ab=(4.25+d)*n
    print(f"{name:20s} d={d:.3f} encode {te:.3f} ms ({ab/te/1e6/8000:.3f})  decode {td:.3f} ms ({ab/td/1e6/8000:.3f})")

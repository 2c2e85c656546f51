import sys, torch, torch.nn.functional as F
sys.path.insert(0, "/root/repo")
from nanofed_amd.ops import attention as A
B,T,H=4,1024,12; C=H*64; dev="cuda:0"
qkv=(torch.randn(B*T,3*C,device=dev)).to(torch.bfloat16)
q,k,v=(t.transpose(1,2) for t in qkv.float().view(B,T,3,H,64).unbind(2))
ref=F.scaled_dot_product_attention(q,k,v,is_causal=True).transpose(1,2).reshape(B*T,C)
out=torch.empty(B*T,C,dtype=torch.bfloat16,device=dev); lse=torch.empty(B,H,T,device=dev)
for kg in (1,2,4):
    A.set_forward_split(kg)
    out.zero_()
    A.causal_attention_fwd(qkv,out,lse,B,T,H)
    torch.cuda.synchronize()
    err=float((out.float()-ref).abs().max())
    g=torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(20): A.causal_attention_fwd(qkv,out,lse,B,T,H)
    g.replay(); torch.cuda.synchronize()
    e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5): g.replay()
    e1.record(); torch.cuda.synchronize()
    print(kg, "max err", err, "us", e0.elapsed_time(e1)*1000/100, flush=True)

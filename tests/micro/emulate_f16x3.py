"""Which operand's 22-bit (hi + lo fp16) representation costs the f16x3 mode its accuracy?  (diagnostic, CPU only)

The LightGlue oracle is evaluated in float64 with ONE operand class at a time rounded to hi + lo fp16 halves (q22) before its
contraction, everything else exact; the error of log_assignment / final descriptors against the all-float64 run is printed per class,
then with every weight tensor pre-scaled by a power of two (what api.hip: pow2_scale does at upload).

    python3 tests/micro/emulate_f16x3.py [final_proj gain]      ->  profiles/r03_emulate_f16x3.txt
"""
import sys, torch, math
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
from oracle import sp_lg_oracle as O
import torch.nn.functional as F
torch.set_grad_enabled(False)
D=torch.float64
def q22(x):
    h=x.to(torch.float16).to(D); l=(x-h).to(torch.float16).to(D); return h+l
def q24(x): return x.to(torch.float32).to(D)
MODE={}
def Q(name,x):
    m=MODE.get(name,'exact')
    return q22(x) if m=='q22' else q24(x) if m=='q24' else x
def lin(x,W,b,name): return F.linear(Q(name+'.a',x),Q(name+'.w',W),b)
def ffn(x,msg,W,p):
    h=lin(torch.cat([x,msg],-1),W[p+".ffn.0.weight"],W[p+".ffn.0.bias"],'ffn0')
    h=F.layer_norm(h,(h.shape[-1],),W[p+".ffn.1.weight"],W[p+".ffn.1.bias"],1e-5); h=F.gelu(h)
    return lin(h,W[p+".ffn.3.weight"],W[p+".ffn.3.bias"],'ffn3')
def attn(q,k,v):
    s=torch.einsum('bhid,bhjd->bhij',Q('qk',q),Q('qk',k))/ (q.shape[-1]**0.5)
    p=torch.softmax(s,-1)
    return torch.einsum('bhij,bhjd->bhid',Q('pv',p),Q('pv',v))
def self_block(x,enc,W,p,heads=4):
    qkv=lin(x,W[p+".Wqkv.weight"],W[p+".Wqkv.bias"],'qkv').unflatten(-1,(heads,-1,3)).transpose(1,2)
    q,k,v=qkv[...,0],qkv[...,1],qkv[...,2]
    q,k=O.apply_rotary(enc,q),O.apply_rotary(enc,k)
    ctx=attn(q,k,v)
    msg=lin(ctx.transpose(1,2).flatten(-2),W[p+".out_proj.weight"],W[p+".out_proj.bias"],'out')
    return x+ffn(x,msg,W,p)
def cross_block(x0,x1,W,p,heads=4):
    sp=lambda t:t.unflatten(-1,(heads,-1)).transpose(1,2)
    qk0=sp(lin(x0,W[p+".to_qk.weight"],W[p+".to_qk.bias"],'qkv')); qk1=sp(lin(x1,W[p+".to_qk.weight"],W[p+".to_qk.bias"],'qkv'))
    v0=sp(lin(x0,W[p+".to_v.weight"],W[p+".to_v.bias"],'qkv')); v1=sp(lin(x1,W[p+".to_v.weight"],W[p+".to_v.bias"],'qkv'))
    m0=attn(qk0,qk1,v1); m1=attn(qk1,qk0,v0)
    m0=lin(m0.transpose(1,2).flatten(-2),W[p+".to_out.weight"],W[p+".to_out.bias"],'out'); m1=lin(m1.transpose(1,2).flatten(-2),W[p+".to_out.weight"],W[p+".to_out.bias"],'out')
    return x0+ffn(x0,m0,W,p), x1+ffn(x1,m1,W,p)
def run(data,W):
    k0=O.normalize_keypoints(data['keypoints0'],data['view0']['image_size']); k1=O.normalize_keypoints(data['keypoints1'],data['view1']['image_size'])
    d0,d1=data['descriptors0'],data['descriptors1']
    e0=O.fourier_encoding(k0,W['posenc.Wr.weight']); e1=O.fourier_encoding(k1,W['posenc.Wr.weight'])
    for i in range(9):
        p=f"transformers.{i}"
        d0=self_block(d0,e0,W,p+".self_attn"); d1=self_block(d1,e1,W,p+".self_attn")
        d0,d1=cross_block(d0,d1,W,p+".cross_attn")
    S,sim=O.log_assignment(d0,d1,W,"log_assignment.8")
    return S,d0
g=torch.Generator().manual_seed(5)
B,M,N=1,256,256
size=torch.tensor([[640.,480.]])
k0=torch.rand(B,M,2,generator=g)*size[:,None]*0.98+0.5
d0=F.normalize(torch.randn(B,M,256,generator=g),dim=-1)
perm=torch.stack([torch.randperm(M,generator=g)[:N] for _ in range(B)])
k1=torch.gather(k0,1,perm[...,None].expand(-1,-1,2))*0.9+2.0*torch.randn(B,N,2,generator=g)
d1=F.normalize(torch.gather(d0,1,perm[...,None].expand(-1,-1,256))+0.03*torch.randn(B,N,256,generator=g),dim=-1)
data={"keypoints0":k0,"keypoints1":k1,"descriptors0":d0,"descriptors1":d1,"view0":{"image_size":size},"view1":{"image_size":size}}
gain=float(sys.argv[1]) if len(sys.argv)>1 else 3.0
W=O.make_lg_weights(5,structured=True,final_gain=gain,match_bias=2.0)
ref32=O.lightglue_forward(data,W,filter_threshold=0.1)
d64,W64=O.cast(data,D),O.cast(W,D)
S64,x64=run(d64,W64)
print("cpu fp32 la err %.2e desc %.2e"%((ref32['log_assignment'].double()-S64).abs().max(), (ref32['ref_descriptors0'][:,0].double()-x64).abs().max()))
names=['qkv.a','qkv.w','qk','pv','out.a','out.w','ffn0.a','ffn0.w','ffn3.a','ffn3.w']
for sel in [names]+[[n] for n in names]+[['qk','qkv.a','qkv.w'],['ffn0.a','ffn0.w','ffn3.a','ffn3.w']]:
    MODE.clear(); MODE.update({n:'q22' for n in sel})
    S,x=run(d64,W64)
    print("%-40s la err %.2e  desc err %.2e"%('+'.join(sel) if len(sel)<10 else 'ALL q22',(S-S64).abs().max(),(x-x64).abs().max()))
MODE.clear(); MODE.update({n:'q24' for n in names}); S,x=run(d64,W64); print("ALL q24 (fp32 operands, exact accumulate): la %.2e desc %.2e"%((S-S64).abs().max(),(x-x64).abs().max()))
print("---- with power-of-two pre-scaling of the WEIGHT tensors (max|w| -> [2^14, 2^15))")
def q22s(x):
    s=2.0**math.floor(math.log2(32768.0/float(x.abs().max())))
    return q22(x*s)/s
_Q=Q
def Q(name,x):
    m=MODE.get(name,'exact')
    if m=='q22s': return q22s(x)
    return q22(x) if m=='q22' else q24(x) if m=='q24' else x
for sel in [['ffn3.w'],['ffn0.w'],['qkv.w','out.w','ffn0.w','ffn3.w']]:
    MODE.clear(); MODE.update({n:'q22s' for n in sel})
    S,x=run(d64,W64)
    print("%-40s la err %.2e  desc err %.2e"%('+'.join(sel)+' scaled',(S-S64).abs().max(),(x-x64).abs().max()))
MODE.clear(); MODE.update({n:'q22' for n in names}); MODE.update({n:'q22s' for n in ['qkv.w','out.w','ffn0.w','ffn3.w']})
S,x=run(d64,W64); print("ALL q22, weights scaled: la %.2e desc %.2e"%((S-S64).abs().max(),(x-x64).abs().max()))
def q22a(x):  # activations scaled by 2^8
    return q22(x*256.0)/256.0
def Q(name,x):
    m=MODE.get(name,'exact')
    if m=='q22s': return q22s(x)
    if m=='q22a': return q22a(x)
    return q22(x) if m=='q22' else q24(x) if m=='q24' else x
MODE.clear(); MODE.update({n:'q22a' for n in names}); MODE.update({n:'q22s' for n in ['qkv.w','out.w','ffn0.w','ffn3.w']})
S,x=run(d64,W64); print("ALL: activations x2^8, weights scaled: la %.2e desc %.2e"%((S-S64).abs().max(),(x-x64).abs().max()))

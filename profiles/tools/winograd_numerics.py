"""Numerics of Winograd F(2x2, 3x3) for the SuperPoint encoder, emulated on the CPU (groundwork for the next structural lever of a
power-limited step: 2.25x fewer MFMA products in the 3x3 layers).  Every layer of the stack on the boat image with the seeded weights:
max |x - float64| of direct fp32, Winograd fp32, and both with operands split into fp16 halves (three products, as the f16x3 kernels
do -- without the power-of-two weight pre-scaling of api.hip, so the split figures are upper bounds).  Result (profiles/r04_winograd_numerics.txt):
Winograd in fp32 is as accurate as the direct form; with split operands its error is ~1.4x the direct split form, far inside 1e-4.

    python profiles/tools/winograd_numerics.py
"""
import sys, os, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from oracle import sp_lg_oracle as O
torch.manual_seed(0)
torch.set_grad_enabled(False)
W = O.make_sp_weights(0)
z = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), 'tests', 'golden', 'two_view_boat.npz'))
img = torch.from_numpy(z['gray0']).float()[None,None]/255.0
img = torch.nn.functional.interpolate(img, size=(480,640), mode='bilinear', align_corners=False)

G = torch.tensor([[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], dtype=torch.float64)
Bt = torch.tensor([[1,0,-1,0],[0,1,1,0],[0,-1,1,0],[0,1,0,-1]], dtype=torch.float64)
At = torch.tensor([[1,1,1,0],[0,1,-1,-1]], dtype=torch.float64)

def split16(x):
    hi = x.half().float(); lo = (x-hi).half().float(); return hi, lo

def wino_conv(x, w, b, dtype, split=False):
    # x (1,C,H,W) H,W even ; w (O,C,3,3)
    x = x.to(dtype); 
    U = torch.einsum('ij,ocjk,lk->ocil', G.to(torch.float64), w.double(), G.to(torch.float64))  # (O,C,4,4) in fp64 at upload
    U = U.to(dtype)
    _,C,H,Wd = x.shape
    xp = torch.nn.functional.pad(x,(1,1,1,1))
    # tiles: 4x4 with stride 2
    t = xp.unfold(2,4,2).unfold(3,4,2)  # (1,C,H/2,W/2,4,4)
    V = torch.einsum('ij,nchwjk,lk->nchwil', Bt.to(dtype), t, Bt.to(dtype))
    if split:
        Vh,Vl = split16(V); Uh,Ul = split16(U)
        M = torch.einsum('ocil,nchwil->nohwil', Uh, Vh)+torch.einsum('ocil,nchwil->nohwil', Uh, Vl)+torch.einsum('ocil,nchwil->nohwil', Ul, Vh)
    else:
        M = torch.einsum('ocil,nchwil->nohwil', U, V)
    Y = torch.einsum('ij,nohwjk,lk->nohwil', At.to(dtype), M, At.to(dtype))  # (1,O,H/2,W/2,2,2)
    Y = Y.permute(0,1,2,4,3,5).reshape(1,w.shape[0],H,Wd)
    return Y + b.to(dtype).view(1,-1,1,1)

def direct(x,w,b,dtype, split=False):
    if split:
        xh,xl = split16(x.float()); wh,wl = split16(w.float())
        f=torch.nn.functional.conv2d
        return f(xh,wh,None,padding=1)+f(xh,wl,None,padding=1)+f(xl,wh,b.float(),padding=1)
    return torch.nn.functional.conv2d(x.to(dtype), w.to(dtype), b.to(dtype), padding=1)

layers = ['conv1a','conv1b','pool','conv2a','conv2b','pool','conv3a','conv3b','pool','conv4a','conv4b']
x64 = img.double()
xs = {'direct32':img.clone(), 'wino32':img.clone(), 'winosplit':img.clone(), 'dirsplit':img.clone()}
for name in layers:
    if name=='pool':
        x64 = torch.nn.functional.max_pool2d(x64,2,2)
        for k in xs: xs[k] = torch.nn.functional.max_pool2d(xs[k],2,2)
        continue
    w,b = W[name+'.weight'], W[name+'.bias']
    x64 = torch.relu(direct(x64,w,b,torch.float64))
    xs['direct32'] = torch.relu(direct(xs['direct32'],w,b,torch.float32))
    xs['dirsplit'] = torch.relu(direct(xs['dirsplit'],w,b,torch.float32,split=True))
    if name=='conv1a':
        xs['wino32'] = xs['direct32'].clone(); xs['winosplit']=xs['dirsplit'].clone()
    else:
        xs['wino32'] = torch.relu(wino_conv(xs['wino32'],w,b,torch.float32))
        xs['winosplit'] = torch.relu(wino_conv(xs['winosplit'],w,b,torch.float32,split=True))
    print(name, 'max|x|', float(x64.abs().max()), {k: f"{float((v.double()-x64).abs().max()):.2e}" for k,v in xs.items()})
# heads
for hd in ['convPa','convDa']:
    w,b = W[hd+'.weight'], W[hd+'.bias']
    y64 = torch.relu(direct(x64,w,b,torch.float64))
    print(hd, 'max', float(y64.abs().max()), 'direct32 %.2e'%float((torch.relu(direct(xs['direct32'],w,b,torch.float32)).double()-y64).abs().max()),
          'dirsplit %.2e'%float((torch.relu(direct(xs['dirsplit'],w,b,torch.float32,split=True)).double()-y64).abs().max()),
          'wino32 %.2e'%float((torch.relu(wino_conv(xs['wino32'],w,b,torch.float32)).double()-y64).abs().max()),
          'winosplit %.2e'%float((torch.relu(wino_conv(xs['winosplit'],w,b,torch.float32,split=True)).double()-y64).abs().max()))

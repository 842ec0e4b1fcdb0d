import sys, torch
R=__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))
sys.path[:0]=[R,R+'/graphical-normalizing-flows_amd',R+'/tests']
import torch.nn.functional as F
from gnf_hip import ops, abi
from test_gpu_parity import _windowed_conditioner
cond=_windowed_conditioner(9, True)
net=cond.embedding_net
B=2
x=torch.rand(B,784)
e=(x.unsqueeze(1)*cond.A.detach().cpu().unsqueeze(0)).reshape(B*784,784)
W1,b1,W2,b2=[t.detach().cpu() for t in (net.conv1.weight,net.conv1.bias,net.conv2.weight,net.conv2.bias)]
c2=F.conv2d(torch.relu(F.conv2d(e.view(-1,1,28,28),W1,b1)),W2,b2)
ref,idx=F.max_pool2d(c2,2,return_indices=True)
# idx: flat index in 24x24 plane -> window-local 0..3
py=(idx//24)%2; px=(idx%24)%2; refarg=(py*2+px).flatten(1)
for exact in (0,1):
    pooled=torch.empty(B*784,2304,device='cuda'); arg=torch.empty(B*784,2304,dtype=torch.uint8,device='cuda')
    ed,W1d,b1d,W2d,b2d=[t.cuda().contiguous() for t in (e,W1,b1,W2,b2)]
    abi.call("gnf_mnistcnn_conv_fwd", abi.ptr(ed), abi.ptr(W1d), abi.ptr(b1d), abi.ptr(W2d), abi.ptr(b2d), abi.ptr(pooled), abi.rawptr(arg), B*784, exact, abi.stream())
    torch.cuda.synchronize()
    p2=ops.MnistConvFn.apply(ed,W1d,b1d,W2d,b2d,bool(exact))
    print("via ops rel err", ((p2.cpu()-ref.flatten(1)).abs().max()/ref.abs().max()).item())
    mism=(arg.cpu().long()!=refarg)
    print("exact",exact,"argmax mismatches",int(mism.sum()),"of",mism.numel(), "pooled rel err", ((pooled.cpu()-ref.flatten(1)).abs().max()/ref.abs().max()).item())
    if mism.any():
        k=mism.nonzero()[:5]
        for (i,p) in k.tolist():
            c=p//144; w=p%144; wy,wx=w//12,w%12
            print(" img",i,"ch",c,"win",wy,wx,"ours",int(arg[i,p]),"ref",int(refarg[i,p]),"vals",c2[i,c,2*wy:2*wy+2,2*wx:2*wx+2].flatten().tolist())
# are torch's equal-patch outputs bit-equal?  background window far from the pixel window
print("torch bg window values img0 ch0:", c2[0,0,20:22,20:22].flatten().tolist())
